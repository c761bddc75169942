"""The final bench line stays small enough to be read: built from a committed sample of the detail dict (round 5's 28 KB line,
tests/golden/bench_detail_sample.json) it must stay under 8 000 bytes, carry the contract's keys, and drop nothing the judge reads."""
import io
import json
import os

import pytest

from gauss_amd import benchline

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def detail():
    with open(os.path.join(HERE, "golden", "bench_detail_sample.json")) as fh:
        return json.load(fh)


def test_final_line_is_small_and_complete(detail):
    s = benchline.final_line(detail)
    assert "\n" not in s
    assert len(s) < benchline.LIMIT < 8000, len(s)
    line = json.loads(s)
    for k in benchline.REQUIRED + ("cpu_baseline", "parity_spot", "launch_form", "result_digest", "emulated_strong_scaling", "end_to_end",
                                   "other_configs"):
        assert k in line, k
    # the contract's figures are the detail's, digit for digit
    assert line["value"] == detail["value"] and line["ms_per_step"] == detail["ms_per_step"]
    assert line["metric"] == detail["metric"] and line["n_gpus"] == 1 and line["vs_baseline"] is None
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert "traffic" in r
    cb = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(cb) and len(cb["sample"]) <= 200
    assert len(line["config"]["workload"]) <= 300
    assert line["end_to_end"]["emulated_world8"]["predicted_efficiency"] == pytest.approx(
        detail["end_to_end"]["emulated_world8"]["predicted_efficiency"], rel=1e-5)
    for k in ("computeLD", "dist", "jepegmix"):
        oc = line["other_configs"][k]
        assert set(("value", "unit", "ms_per_step", "roofline", "cpu_baseline", "parity_ok")) <= set(oc)
    assert line["other_configs"]["computeLD"]["one_resident_window"]["gram_tflops"] > 0


def test_emit_prints_one_stdout_line_and_the_detail_elsewhere(detail, tmp_path, monkeypatch):
    """stdout: the ONE line; stderr: one short pointer (the driver's record is the tail of stdout FOLLOWED by stderr, so a bulky
    stderr would push the line out of it); the detail: the file, and a copy in every also_dir."""
    monkeypatch.delenv("GAUSS_BENCH_DETAIL_STDERR", raising=False)
    out, err = io.StringIO(), io.StringIO()
    path = str(tmp_path / "bench_detail.json")
    side = str(tmp_path / "gpurun_out")
    line = benchline.emit(detail, headline=True, detail_path=path, stdout=out, stderr=err, also_dirs=(side,))
    lines = [l for l in out.getvalue().split("\n") if l]
    assert len(lines) == 1 and json.loads(lines[0]) == line and len(lines[0]) < benchline.LIMIT
    assert len(err.getvalue()) < 400 and "bench_detail.json" in err.getvalue()
    assert len(out.getvalue()) + len(err.getvalue()) < 7000
    for p in (path, os.path.join(side, "bench_detail.json")):
        with open(p) as fh:
            assert json.load(fh)["value"] == detail["value"]
    assert line["detail"] == "bench_detail.json"
    # on request the detail is printed inline, on stderr, tagged
    monkeypatch.setenv("GAUSS_BENCH_DETAIL_STDERR", "1")
    out, err = io.StringIO(), io.StringIO()
    benchline.emit(detail, headline=True, detail_path=path, stdout=out, stderr=err)
    tagged = json.loads(err.getvalue())
    assert list(tagged) == ["detail"] and tagged["detail"]["value"] == detail["value"]
    assert len([l for l in out.getvalue().split("\n") if l]) == 1


def test_a_block_that_outgrows_the_line_is_refused(detail):
    d = dict(detail, config=dict(detail["config"], windows_per_rank=list(range(4000))))
    with pytest.raises(ValueError):
        benchline.final_line(d)


def test_other_modes_shrink(detail):
    e2e = {"metric": "m", "value": 1.0, "unit": "u", "n_gpus": 1, "steps": 1, "warmup": 1, "ms_per_step": 1.0, "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "config": {"workload": "x" * 1000},
           "roofline": None, "end_to_end": detail["end_to_end"]}
    s = benchline.final_line(e2e, headline=False)
    assert len(s) < 8000
    assert json.loads(s)["value"] == 1.0
