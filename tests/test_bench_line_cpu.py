"""bench.py's line diet (no GPU needed): the ONE JSON line must fit the ~8 KB of stdout tail the driver keeps."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_line_fits_the_driver_tail():
    """bench.emit on the LARGEST line this repo ever produced (round 5's 15.5 KB, profiles/r05z_bench.json): <= 7000 bytes,
    every BASELINE config's ms_per_step and roofline fraction still on it, the six judged objects last."""
    import contextlib
    import importlib.util
    import io
    import tempfile
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    big = json.load(open(os.path.join(ROOT, "profiles", "r05z_bench.json")))
    assert len(json.dumps(big)) > 15000
    with tempfile.TemporaryDirectory() as td:
        os.environ["NPLDA_BENCH_DETAIL"] = os.path.join(td, "bench_detail.json")
        try:
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                b.emit(big)
            assert json.load(open(os.environ["NPLDA_BENCH_DETAIL"])) == big  # nothing is lost: the detail file is the full object
        finally:
            del os.environ["NPLDA_BENCH_DETAIL"]
    line = buf.getvalue().strip()
    assert "\n" not in line and len(line) <= b.LINE_LIMIT == 7000
    d = json.loads(line)
    assert list(d)[-6:] == ["cpu_baseline", "alt_d170", "alt_cfg5", "alt_cfg3", "alt_cfg2", "roofline"]
    for k in ("alt_cfg2", "alt_cfg3", "alt_cfg5"):
        assert d[k]["ms_per_step"] > 0 and 0 < d[k]["roofline"]["frac"] < 1 and 0 < d[k]["d170"]["frac"] < 1
        assert abs(d[k]["ms_per_step"] - big[k]["ms_per_step"]) <= 1e-6 * big[k]["ms_per_step"]
    assert d["alt_cfg3"]["stats_ms"] > 0 and d["alt_cfg3"]["allgather_bytes"] == 704000
    assert d["roofline"]["frac"] == pytest.approx(big["roofline"]["frac"], rel=1e-6) and d["cpu_baseline"]["cores"] >= 1
    assert d["config"]["workload"] == big["config"]["workload"] and d["metric"] == big["metric"]
