bash tools/gpu_visit.sh "attention or config4 or config5"
for round in 1 2; do
  echo -n "fwd DMA (default)    "; python tools/attn_bench.py 2>/dev/null | tr '\n' ' '; echo
  echo -n "fwd register-staged  "; UVTG_ATTN_FWD_DMA_OFF=1 python tools/attn_bench.py 2>/dev/null | tr '\n' ' '; echo
done | tee gpurun_out/visit_attn_bench_fwd.txt
AB_ARGS="--config 4" bash tools/ab5.sh 2 "c4 fwd register-staged|UVTG_ATTN_FWD_DMA_OFF=1" "c4 fwd DMA (default)|" 2>&1 | tee gpurun_out/visit_ab_fwd_dma.txt
AB_ARGS="--config 5" bash tools/ab5.sh 1 "c5 fwd register-staged|UVTG_ATTN_FWD_DMA_OFF=1" "c5 fwd DMA (default)|" 2>&1 | tee -a gpurun_out/visit_ab_fwd_dma.txt
