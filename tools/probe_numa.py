"""Does the host-fed rate depend on which NUMA node the process runs on?  Reads the GPU's node and local CPU list from sysfs, then runs
tools/host_fed.py's graph rows in subprocesses bound (os.sched_setaffinity before torch is imported) to the GPU-local CPUs, to the other
CPUs, and unbound.    python tools/probe_numa.py [--out gpurun_out/numa_host_fed.json]"""
import argparse, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_cpulist(s):
    cpus = set()
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_nodes():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        try:
            vendor = open(os.path.join(dev, "vendor")).read().strip()
            if vendor != "0x1002":
                continue
            out.append({"device": os.path.realpath(dev), "numa_node": int(open(os.path.join(dev, "numa_node")).read()),
                        "local_cpulist": open(os.path.join(dev, "local_cpulist")).read().strip()})
        except OSError:
            pass
    return out


CHILD = r'''
import os, sys, json
cpus = json.loads(sys.argv[1])
if cpus: os.sched_setaffinity(0, cpus)
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, os.path.join(sys.argv[2], "tools"))
import host_fed
r = host_fed.measure(16, reps=3, warmup=2)
print("RESULT " + json.dumps([(x["node"][:40], x["input"], x["Mpix_s"]) for x in r["rows"]]))
'''

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/numa_host_fed.json")
    a = ap.parse_args()
    gpus = gpu_nodes()
    allc = os.sched_getaffinity(0)
    print("[numa] gpus", gpus, flush=True)
    print("[numa] affinity of this process:", len(allc), "cpus", flush=True)
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node*")):
        nodes[os.path.basename(d)] = open(os.path.join(d, "cpulist")).read().strip()
    print("[numa] nodes", nodes, flush=True)
    res = {"gpus": gpus, "nodes": nodes, "affinity_cpus": len(allc), "runs": {}}
    local = (parse_cpulist(gpus[0]["local_cpulist"]) & allc) if gpus else set()
    cases = {"unbound": []}
    if local and local != allc:
        cases["gpu_local_cpus"] = sorted(local)
        cases["other_cpus"] = sorted(allc - local)
    for name, cpus in cases.items():
        p = subprocess.run([sys.executable, "-c", CHILD, json.dumps(cpus), ROOT], capture_output=True, text=True, timeout=400)
        line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")]
        res["runs"][name] = json.loads(line[0][7:]) if line else {"error": p.stderr[-400:]}
        print("[numa]", name, len(cpus), "cpus:", res["runs"][name], flush=True)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)
