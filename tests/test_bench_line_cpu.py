"""bench.py prints ONE final stdout line the driver can parse from an 8 KB tail: the compact line is built from a canned
full result (tests/golden/bench_full_canned.json: a round-4 run with thirteen legs, 34 KB) and must stay under 8 KB with the headline
objects intact."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _canned():
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_canned.json")))
    # what this round's bench adds to a full result
    d["config"]["gemm_mode"] = "bf16x6"
    d["f32_mfma_twin"] = {"what": "x" * 200, "value": 5.2e7, "unit": "env-steps/s", "ms_per_step": 9.4, "dtype": "f32", "last_loss": 0.2,
                          "learner_updates_per_sec": 134.0, "roofline": dict(d["roofline"])}
    for c in d["configs"]:
        c["segments_updates_per_sec"] = [c["learner_updates_per_sec"]] * 3
    for r in (d["roofline"], d["f32_mfma_twin"]["roofline"]):      # round 6: what the matrix cores execute, by model and by counter
        for k in r.get("kernels", []):
            k.update(mfma_flop=2.466e11, mfma_instr_model={"k32": 80000000, "k16": 18432000}, mfma_instr_pmc=99532800.0, mfma_busy_frac=0.45)
    return d


def test_compact_line_fits_the_drivers_tail_and_keeps_the_headline_objects():
    import bench
    full = _canned()
    assert len(json.dumps(full)) > 30000
    txt = bench.compact_line(full)
    assert len(txt) < 8192 and "\n" not in txt
    line = json.loads(txt)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_update", "cpu_baseline", "configs", "f32_mfma_twin"):
        assert k in line, k
    assert line["value"] == float("%.6g" % full["value"]) and line["config"]["workload"] == "qmix_2s3z_T120_envs4096"
    r = line["roofline"]
    assert r["frac"] > 0 and r["bound"] == "mfma" and r["peak"] > 0 and r["traffic"] > 0 and r["unit"] == "TFLOP/s"
    assert 1 <= len(r["kernels"]) <= 6 and all({"name", "ms", "frac"} <= set(k) for k in r["kernels"])
    cb = line["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 16 and cb["kind"] == "port" and cb["unit"] == "env-steps/s" and cb["sample"]
    assert len(line["configs"]) == len(full["configs"])
    for leg in line["configs"]:
        assert {"workload", "gemm_mode", "updates_per_sec", "roofline_update_frac", "roofline"} <= set(leg)
        assert set(leg["roofline"]) == {"kernel", "frac", "traffic"}
    tw = line["f32_mfma_twin"]
    assert tw["value"] > 0 and tw["ms_per_step"] > 0 and tw["roofline"]["frac"] > 0


def test_compact_line_drops_detail_rather_than_overflow():
    import bench
    full = _canned()
    full["configs"] = full["configs"] * 4            # 52 legs: more than 8 KB even in short form
    txt = bench.compact_line(full)
    line = json.loads(txt)
    assert len(txt) <= bench.LINE_LIMIT and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_emit_prints_the_compact_line_last_and_writes_the_full_result(tmp_path, monkeypatch, capsys):
    import bench
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(_canned())
    cap = capsys.readouterr()
    assert len(cap.out.strip().splitlines()) == 1           # stdout: the line and nothing else
    last = cap.out.strip().splitlines()[-1]
    assert len(last) < 8192 and json.loads(last)["full"] == "bench_full.json"
    assert "[bench full]" in cap.err
    full = json.load(open(tmp_path / "bench_full.json"))
    assert len(full["configs"]) == 13 and "kernels" in full["configs"][0]
