"""bench.py's FLOP accounting (host arithmetic only): the encoder section's EXECUTED work must follow what the engine really launches --
SURVEY 8d's algorithmic count, minus the last layer's text rows (engine.hip, last_layer_clip), plus the conv-head weight gradients that ride in
the section's deferred weight-gradient launch on a single rank -- and every experiment switch that changes the launches must change the count."""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(monkeypatch, env):
    for k in ("UVTG_LAST_CLIP_OFF", "UVTG_TN_CONV_DEFER_OFF", "UVTG_TN_DEFER_OFF", "WORLD_SIZE", "UVTG_TN_EVENTS_PER_LAYER"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    sys.path.insert(0, ROOT)
    return importlib.import_module("bench")


def test_encoder_section_flops_follow_the_launches(monkeypatch):
    bench = _bench(monkeypatch, {})
    d, F, E = bench.MODEL["d"], bench.MODEL["F"], bench.MODEL["E"]
    B, Lv, Lt = 256, 75, 32
    S = Lv + Lt
    alg = 3 * E * B * (8 * S * d * d + 4 * S * d * F + 4 * S * S * d)
    clip = 3 * 4 * d * F * B * Lt                      # fwd + dgrad + wgrad of linear1 and linear2 on the last layer's text rows
    conv = 4 * 2 * d * 3 * d * B * (Lv + 2)            # four Conv1d(k=3) weight gradients over the zero-framed rows
    a, e = bench.encoder_flops(None, B, Lv, Lt, False)
    assert a == alg and e == alg - clip + conv
    assert abs(alg / 1e12 - 4.280) < 1e-3 and abs(e / 1e12 - 4.673) < 1e-3      # the numbers DESIGN.md quotes
    monkeypatch.setenv("UVTG_LAST_CLIP_OFF", "1")
    assert bench.encoder_flops(None, B, Lv, Lt, False)[1] == alg + conv
    monkeypatch.setenv("UVTG_TN_CONV_DEFER_OFF", "1")
    assert bench.encoder_flops(None, B, Lv, Lt, False)[1] == alg
    monkeypatch.delenv("UVTG_LAST_CLIP_OFF"); monkeypatch.delenv("UVTG_TN_CONV_DEFER_OFF")
    monkeypatch.setenv("WORLD_SIZE", "8")              # N > 1 (round 6): grouped deferral under the readiness events, the conv gradients ride along
    assert bench.encoder_flops(None, B, Lv, Lt, False)[1] == alg - clip + conv
    monkeypatch.setenv("UVTG_TN_EVENTS_PER_LAYER", "1")   # the per-layer-event mode of rounds 2-5: the conv gradients keep their own launch
    assert bench.encoder_flops(None, B, Lv, Lt, False)[1] == alg - clip


def test_packed_stream_counts_only_the_rows_it_runs(monkeypatch):
    bench = _bench(monkeypatch, {})
    d, F, E = bench.MODEL["d"], bench.MODEL["F"], bench.MODEL["E"]
    B, Lv, Lt = 2, 75, 32
    lens = [([10, 75], [4, 32])]                        # one batch: (clips per sample, text tokens per sample)
    rows = [min(Lv, 10 + 3) + 4, 75 + 32]               # valid clips + 3-clip conv halo + valid text
    want = sum(3 * E * (r * (8 * d * d + 4 * d * F) + 4 * r * r * d) for r in rows) - 3 * 4 * d * F * (4 + 32) + 4 * 2 * d * 3 * d * ((13 + 2) + (75 + 2))
    a, e = bench.encoder_flops(lens, B, Lv, Lt, True)
    assert e == want and a == 3 * E * B * (8 * (Lv + Lt) * d * d + 4 * (Lv + Lt) * d * F + 4 * (Lv + Lt) ** 2 * d)
