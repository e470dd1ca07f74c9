"""bench.py's contract with the driver, as far as it can be checked without a GPU: the ONE stdout line of the default run must fit the driver's 8 KB tail
with every op's headline figures in it (compact()), and carry the fields the judge reads.  Input: the complete object of the last measured default run
(profiles/r06_bench_default_full.json, written by bench.py itself)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


def test_default_line_fits_the_drivers_tail_and_keeps_every_ops_figures():
    bench = load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_default_full.json")))
    line = json.dumps(bench.compact(full))
    assert len(line) < 7700, len(line)                                   # 8 KB tail, with room for longer numbers
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"] and "model" not in d["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    assert d["cpu_baseline"]["kind"] in ("reference", "port")
    extra = d["extra"]
    for op in ("bgzf_deflate", "end_to_end", "cram_rans_nx16_decode", "cram_rans_4x16_decode", "cram_slices", "cram_records_to_bam", "cram_records_encode", "cram31_file_encode", "cram_fqzcomp"):
        assert op in extra, op
        if op != "end_to_end":
            assert isinstance(extra[op].get("value"), (int, float)) and "error" not in extra[op], (op, extra[op])
            assert "frac" in extra[op]["roofline"], op
    assert extra["end_to_end"]["gpu"]["read_GBps"] > 0 and extra["end_to_end"]["reference"]["read_GBps"] > 0
    # the libhts-level figures (the reference's test_view on libhts_gpu.so vs on the reference's libhts) must reach the driver: round 5's line lost them
    view = extra["end_to_end"]["libhts_view"]
    for leg in ("decode", "bam2bam"):
        for side in ("libhts_gpu", "reference"):
            assert view[leg][side]["seconds"] > 0 and view[leg][side]["plain_GBps"] > 0, (leg, side, view)
    # ... and the libhts-level CRAM legs (cram_get_bam_seq / cram_put_bam_seq = the whole-slice reader / writer; *_blocks = the per-block form; *_large = 10.24 M records)
    for leg in ("cram_decode", "cram_encode", "cram31_decode", "cram31_encode", "cram_decode_large", "cram_to_bam_large", "cram_encode_large"):
        assert view[leg]["gpu_s"] > 0 and view[leg]["ref_s"] > 0, (leg, view)
    assert view["cram_decode_blocks"]["gpu_s"] > 0 and view["cram_encode_blocks"]["gpu_s"] > 0


def test_default_arguments_are_one_gpu_and_minutes():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'add_argument("--gpus", type=int, default=1)' in src and 'add_argument("--steps", type=int, default=5)' in src and 'add_argument("--warmup", type=int, default=1)' in src
