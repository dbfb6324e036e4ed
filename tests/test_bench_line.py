"""The bench contract the driver depends on: the LAST stdout line is one JSON object that fits the driver's ~8 KB stdout
tail (round 3's 31.8 KB line left BENCH_r03.json unparsed), and `python bench.py --gpus N` starts its own ranks."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline")


def _bench():
    spec = importlib.util.spec_from_file_location("osn_bench", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("record", ["r03_s17_bench_final.json", "r03_s20_bench_head.json"])
def test_headline_of_a_full_record_fits_the_drivers_tail(record):
    """Round 3's archived full records (31.8 KB each: 147 per-stage dicts, per-kernel survey, prose) through `headline()`."""
    b = _bench()
    detail = json.loads(open(os.path.join(ROOT, "profiles", record)).read().strip().splitlines()[-1])
    line = json.dumps(b.headline(detail, os.path.join(ROOT, "bench_detail.json")), separators=(",", ":"))
    assert len(line) < 8192 and len(line) <= b.MAX_LINE_BYTES
    back = json.loads(line)
    for k in REQUIRED:
        assert k in back, k
    assert back["value"] == detail["value"] and back["ms_per_step"] == detail["ms_per_step"]
    rf = back["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"]
    assert "workload" in back["config"] and "model" not in back["config"]
    assert "stages" not in back and "kernels" not in back
    assert back["detail"] == "bench_detail.json"


def test_headline_drops_side_blocks_rather_than_outgrow_the_limit():
    b = _bench()
    detail = json.loads(open(os.path.join(ROOT, "profiles", "r03_s20_bench_head.json")).read().strip().splitlines()[-1])
    detail["phases"] = {"phase_%04d" % i: {"ms": 1.0 / (i + 1)} for i in range(600)}
    line = json.dumps(b.headline(detail, None), separators=(",", ":"))
    assert len(line) <= b.MAX_LINE_BYTES
    back = json.loads(line)
    assert "phases_ms" not in back and all(k in back for k in REQUIRED)


def test_headline_carries_both_scaling_definitions():
    """at N = 1 the reference's own 1-GPU configuration (8 scenes per step) per scene at top level; at N > 1 the
    one-GPU batch of N scenes and the speed-up against it (run/distill.py:146 `batch_size //= ngpus`) beside the weak-scaling value."""
    b = _bench()
    detail = json.loads(open(os.path.join(ROOT, "profiles", "r03_s20_bench_head.json")).read().strip().splitlines()[-1])
    detail.setdefault("phases", {})["batch8_step"] = {"ms": 43.12, "voxels": 809000, "scenes": 8}
    line = b.headline(detail, None)
    assert abs(line["batch8_per_scene_ms"] - 43.12 / 8) < 1e-3 and "speedup_vs_same_batch_on_one_gpu" not in line
    detail["n_gpus"] = 8
    detail["scaling_reference"] = {"one_gpu_batch_of_n_scenes_ms": 43.0, "n_scenes": 8, "speedup_vs_same_batch_on_one_gpu": 4.8}
    line = b.headline(detail, None)
    assert line["speedup_vs_same_batch_on_one_gpu"] == 4.8 and line["one_gpu_batch_of_n_scenes_ms"] == 43.0
    assert len(json.dumps(line, separators=(",", ":"))) <= b.MAX_LINE_BYTES


def test_headline_carries_the_round5_fields():
    """VERDICT r4 'next' #4: the line says what SURVEY 8(d) defines -- roofline fraction over ALL launches of the dominant shape with
    the forward-pass-only figure beside it, the step with the reference's call sites unchanged next to ms_per_step, the CPU
    baseline's visible cores next to the threads it used, the torch call-site query time next to the kernel's."""
    b = _bench()
    detail = json.loads(open(os.path.join(ROOT, "profiles", "r03_s20_bench_head.json")).read().strip().splitlines()[-1])
    rf = detail["roofline"]
    rf.update({"frac_all": rf["frac"], "frac_fwd": rf["frac"] * 1.06, "avg_launch_us_fwd": rf["avg_launch_us"] / 1.06})
    detail["cpu_baseline"].update({"cores_visible": 256, "cores": 16})
    detail["phases"] = dict(detail.get("phases") or {}, drop_in_step={"ms": 11.24, "voxels_per_s": 100999 / 11.24e-3})
    detail["query"]["torch_call_site"] = {"ms": 0.31}
    back = json.loads(json.dumps(b.headline(detail, None), separators=(",", ":")))
    assert back["roofline"]["frac_all"] == back["roofline"]["frac"] and back["roofline"]["frac_fwd"] > back["roofline"]["frac"]
    assert back["ms_per_step_call_sites_unchanged"] == 11.24 and back["value_call_sites_unchanged"] > 0
    assert back["cpu_baseline"]["cores_visible"] == 256 and back["cpu_baseline"]["cores"] == 16
    assert back["query"]["torch_call_site_ms"] == 0.31


def test_gpus_n_without_a_launcher_spawns_its_own_ranks():
    """`python bench.py --gpus 2` (no torchrun, no RANK in the environment) re-executes under torch.distributed.run; the
    hidden self-test mode runs the N > 1 harness (barrier-bracketed timing, MAX over ranks, rank-0 line) on gloo / CPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--spawn-self-test"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()        # file descriptor 1 carries the headline and NOTHING else (claim_stdout)
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["value"] > 0 and d["config"]["parallelism"] == "dp2"
    assert "re-executing under torch.distributed.run" in r.stderr


def test_spawn_command_is_the_drivers_launch_line():
    b = _bench()
    cmd = b.spawn_command(4, ["--gpus", "4", "--steps", "20"], port=29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "20"]


def test_library_chatter_on_fd1_does_not_reach_the_callers_stdout():
    """RCCL prints its version banner to C stdout; after claim_stdout() that descriptor is stderr and the headline is the only
    thing the caller's stdout sees."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'RCCL version : x\\n'); print('python chatter'); bench.emit_line('{\"a\":1}')") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == '{"a":1}\n'
    assert "RCCL version" in r.stderr and "python chatter" in r.stderr
