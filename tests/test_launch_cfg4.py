"""BASELINE cfg4 launcher (tools/launch_cfg4.sh): N independent processes, one per GPU, pinned with HIP_VISIBLE_DEVICES,
no collectives.  CPU-side check of the launcher itself with a stand-in command (the GPU dry run is in profiles/)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_launcher_pins_one_device_per_job_and_sums_the_lines(tmp_path):
    out = tmp_path / "cfg4"
    # every job prints the device it was given and a bench-shaped JSON line
    code = ("import os, json; print(json.dumps(dict(value=100.0 + {i}, unit='iters/s', ms_per_step=1.0, "
            "dev=os.environ.get('HIP_VISIBLE_DEVICES'), world=os.environ.get('WORLD_SIZE'))))")
    res = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_cfg4.sh"), "-n", "3", "-o", str(out), "--",
                          sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [json.loads(open(out / f"job{i}.out").read().strip().splitlines()[-1]) for i in range(3)]
    assert [l["value"] for l in lines] == [100.0, 101.0, 102.0]
    assert all(l["world"] == "1" for l in lines)                       # independent replicas, no process group
    assert all(l["dev"] is not None and l["dev"].isdigit() for l in lines)
    summary = json.loads(res.stdout.strip().splitlines()[-1])
    assert summary == {"cfg4_jobs": 3, "jobs_reporting": 3, "aggregate_value": 303.0, "scaling": "replicas only"}


def test_launcher_reports_how_many_jobs_are_in_the_aggregate_and_fails_when_one_is_missing(tmp_path):
    out = tmp_path / "cfg4"
    # job 1 prints no bench line: the aggregate must say 2 of 3 and the launcher must not exit 0
    code = ("import json, sys; i = {i}; "
            "print(json.dumps(dict(value=10.0, unit='iters/s', ms_per_step=1.0))) if i != 1 else sys.exit(0)")
    res = subprocess.run(["bash", os.path.join(ROOT, "tools", "launch_cfg4.sh"), "-n", "3", "-o", str(out), "--",
                          sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    summary = json.loads(res.stdout.strip().splitlines()[-1])
    assert summary["cfg4_jobs"] == 3 and summary["jobs_reporting"] == 2 and summary["aggregate_value"] == 20.0
    assert res.returncode != 0
