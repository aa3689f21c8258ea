"""bench.py's output contract, checked without a GPU: argument defaults, the PMC-traffic lookup, and the keys / consistency of
the last committed bench line (profiles/*_bench_line.json is written by `python bench.py` on the MI355X box)."""
import glob
import importlib.util
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("nvsf_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_defaults_and_traffic_lookup(monkeypatch):
    b = _bench()
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = b.parse()
    assert a.gpus == 1 and a.steps >= 20 and a.warmup >= 5 and a.num_rays == 4096 and a.num_rays_lidar == 4096 and a.num_steps == 768
    assert b.HBM_PEAK_GBS == 8000.0 and b.MFMA_PEAK_TFLOPS == 2500.0
    # PMC traffic is only reported from a profile taken on the running kernel sources (no stale constants)
    import glob
    import build as nvsf_build
    prof = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]))
    t = b.pmc_traffic("render_uniform[lidar]")
    if prof.get("csrc_digest") == nvsf_build.csrc_digest():
        assert t["traffic"] is not None and t["traffic"] > 1e8 and "profiles/" in t["traffic_source"]
    else:
        assert t["traffic"] is None and "traffic_stale" in t
    assert b.pmc_traffic("no_such_kernel[lidar]")["traffic"] is None


def test_gpus_flag_starts_that_many_ranks(monkeypatch):
    """`python bench.py --gpus N` without a launcher must start N ranks itself (the flag was parsed and ignored)."""
    import subprocess
    b = _bench()
    cmd = b.launch_command(4, ["--gpus", "4", "--steps", "3"], 29999)
    assert "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    # main(): no WORLD_SIZE + --gpus 3 -> self_launch (and nothing of this process touches the GPU before it)
    seen = {}
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "3"])
    monkeypatch.setattr(b, "self_launch", lambda a: seen.setdefault("gpus", a.gpus) and 0)
    monkeypatch.setattr(b.torch.cuda, "is_available", lambda: (_ for _ in ()).throw(AssertionError("HIP touched before the launch")))
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert seen == {"gpus": 3}
    # a launcher whose world size disagrees with --gpus is an error, not a silent 1-rank run
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    try:
        b.main()
        raise AssertionError("mismatch accepted")
    except SystemExit as e:
        assert "WORLD_SIZE=2" in str(e.code)
    # end to end on a box without a GPU: two ranks come up, each refuses to run without a HIP device, the parent fails loudly
    import torch
    if torch.cuda.device_count() == 0:
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                           text=True, env=env, timeout=300)
        # (the launcher stops the other rank as soon as the first one fails: it either printed the same refusal or shows up in
        # the launcher's failure report as rank 1)
        assert r.returncode != 0 and "needs a HIP device" in r.stderr, r.stderr[-2000:]
        assert r.stderr.count("needs a HIP device") >= 2 or re.search(r"rank\s*: 1 \(local_rank: 1\)", r.stderr), r.stderr[-2000:]
        assert not r.stdout.strip()


def test_committed_bench_line_follows_the_contract():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_line.json")))
    assert files
    text = open(files[-1]).read().strip().splitlines()[-1]
    # the ONE stdout line must stay small enough for the driver to parse (BENCH_r04.parsed was null at 29.7 KB)
    b = _bench()
    assert len(text) <= b.FINAL_LINE_MAX_BYTES <= 8192, len(text)
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "rays/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    rays = (d["config"]["num_rays"] + d["config"]["num_rays_lidar"]) * d["n_gpus"]
    assert abs(d["value"] - rays / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3  # value = whole-job rays / time (5 significant digits)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic bytes per launch / live launch duration
    if r["bound"] == "hbm":
        per_unit = float(r["algorithmic"].split()[0])
        assert abs(r["achieved"] - per_unit * r["units_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) / r["achieved"] < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "rays/s" and c["sample"]
    # one-number summaries of the legs; the rows they summarise are in the detail file committed beside the line
    for leg in ("occupancy", "dynamic", "train", "raymarching", "field_ops", "eval", "reference_default_grid", "outputs_match_oracle"):
        assert leg in d, leg
    assert d["outputs_match_oracle"]["ok"] is True and d["train"]["grads_match"]["ok"] is True
    assert d["dynamic"]["outputs_match_fixture"]["ok"] is True
    occ = d["occupancy"]["outputs_match_oracle"]  # config 3 carries a parity field of its own
    assert occ["ok"] is True and occ["tolerance"] == 1e-4 and min(occ["checked_rays"]) >= 64 and min(occ["exact_fraction"]) >= 0.97
    det = json.load(open(files[-1].replace("_bench_line.json", "_bench_detail.json")))
    assert abs(det["value"] - d["value"]) / d["value"] < 1e-3
    for row in det["raymarching"]["kernels"] + det["field_ops"]["kernels"] + det["kernels"]:
        assert row["ms"] > 0 and row["frac"] > 0 and row["bound"] in ("hbm", "mfma", "l1")  # l1: L2-resident gathers (secondary rows only)
    for row in det["field_ops"]["kernels"]:  # stand-alone MLP rows keep BOTH columns
        if row["kernel"].startswith("mlp_"):
            assert row["bound"] == "hbm" and 0 < row["mfma_frac"] < 1


def test_compact_line_stays_small_whatever_the_legs_return():
    """compact_line() on the last committed detail reproduces a line under the limit, and sheds summaries rather than the contract
    keys when a leg grows."""
    b = _bench()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_detail.json")))
    assert files
    det = json.load(open(files[-1]))
    line = b.compact_line(det, "gpurun_out/bench_detail.json")
    assert len(json.dumps(line)) <= b.FINAL_LINE_MAX_BYTES and "shed" not in line
    det["train"]["kernels"]["rows"] = [dict(r, kernel="k" * 28 + str(i)) for i, r in enumerate(det["train"]["kernels"]["rows"] * 40) if r.get("bound") == "mfma"]
    line = b.compact_line(det, None)
    assert len(json.dumps(line)) <= b.FINAL_LINE_MAX_BYTES
    for k in ("metric", "value", "roofline", "cpu_baseline", "config"):
        assert k in line


def test_compact_line_survives_skipped_legs():
    """N > 1 ranks, --no-extra-legs or a run under rocprofv3 leave legs out or reduce them to {"skipped": ...}: the summary must not
    raise between the timed loop and the line (the first r05 two-rank run died there with a KeyError)."""
    b = _bench()
    det = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_detail.json")))[-1]))
    thin = {k: det[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                                "data", "config")}
    thin["train"] = {"value": 1.0, "ms_per_step": 2.0, "steps": 3, "kernels": {"skipped": "running under rocprofv3"}, "grads_match": None,
                     "allreduce": {"payload_MB": 1.0, "buckets": 2, "per_bucket": []}, "allreduce_collectives_per_step": 3}
    thin["dynamic"] = {"value": 1.0}
    thin["occupancy"] = {}
    thin["raymarching"] = {"error": "x"}
    line = b.compact_line(thin, None)
    assert line["train"]["ms_per_step"] == 2.0 and "launches_per_step" not in line["train"] and line["train"]["allreduce"]["buckets"] == 2
    assert line["dynamic"]["train"] == {"value": None, "ms_per_step": None} and len(json.dumps(line)) <= b.FINAL_LINE_MAX_BYTES


def test_two_rank_line_carries_what_a_scaling_run_is_checked_by():
    """The N > 1 control flow of bench.py as it ran on a 1-GPU box (two ranks sharing cuda:0 over gloo, NVSF_BENCH_SAME_DEVICE=1, line
    tagged `invalid`): the fields the first real 8-GPU run will be checked by must be there -- every rank loaded the same library
    (version, digest of all kernel sources, sha1 of the shared object), how many ranks each rank saw, every rank's own step time,
    and for the training leg the gradient all-reduce's payload, bucket times and the ring estimate it is to be compared with."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]*_bench2_same_device_line.json")))
    assert files, "no round-3 two-rank line committed"
    d = json.loads(open(files[-1]).read().strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and "invalid" in d
    if "detail" in d:  # round 5 on: the stdout line is the compact one, the per-bucket rows are in the detail file committed beside it
        assert len(open(files[-1]).read().strip()) <= _bench().FINAL_LINE_MAX_BYTES
        assert d["train"]["allreduce"]["buckets"] >= 2 and d["train"]["allreduce_collectives_per_step"] >= 2
        det = json.load(open(files[-1].replace("_line.json", "_detail.json")))
        assert det["n_gpus"] == 2 and abs(det["value"] - d["value"]) / d["value"] < 1e-3
        d = det
    assert d["ranks_seen"] == {"min": 2, "max": 2, "answered": 2}
    assert d["build"]["ranks_agree"] is True and len(d["build"]["csrc_digest"]) == 16 and d["build"]["version"].startswith("nvsf_hip")
    assert len(d["per_rank_ms_per_step"]) == 2 and max(d["per_rank_ms_per_step"]) == pytest_approx(d["ms_per_step"])
    tr = d["train"]
    assert len(tr["per_rank_ms_per_step"]) == 2 and tr["allreduce_collectives_per_step"] >= 2
    ar = tr["allreduce"]
    assert ar["payload_MB"] > 90 and ar["buckets"] == len(ar["per_bucket"]) and ar["sum_ms"] > 0  # two 24.4 M-entry tables + MLPs, fp32
    assert abs(sum(b["MB"] for b in ar["per_bucket"]) - ar["payload_MB"]) < 1e-6
    assert ar["ring_estimate_ms"] == pytest_approx(2.0 * (2 - 1) / 2 * ar["payload_MB"] * 1e6 / 153e9 * 1e3)


def pytest_approx(v, rel=1e-6):
    import pytest
    return pytest.approx(v, rel=rel)
