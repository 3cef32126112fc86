bash tools/gpu_visit.sh "last_layer or delta_from or projection_mode_flip or nt256_engine or survives or bench_path_trainstep or packed or ragged or graph"
bash tools/ab5.sh 2 "A every row|UVTG_LAST_CLIP_OFF=1" "A clip rows (default)|" "A clip rows, short groups last|UVTG_TN_SHORT_LAST=1" 2>&1 | tee gpurun_out/visit_ab_clip3.txt
