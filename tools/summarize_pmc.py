"""Summarise rocprofv3 counter_collection + kernel_trace CSVs under the given directories: per kernel name the average
of every counter and the average duration; for GRBM_GUI_ACTIVE also the effective clock (counter / duration)."""
import csv, glob, os, sys, collections, json

def load(d):
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    durs = collections.defaultdict(list)
    per_dispatch = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "vrg" not in name: continue
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            durs[(os.path.relpath(os.path.dirname(f), d).split(os.sep)[0], name[:70])].append(dur)
            per_dispatch[(f.replace("kernel_trace", "X"), r.get("Dispatch_Id"))] = dur
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "")
            if "vrg" not in name: continue
            key = (os.path.relpath(os.path.dirname(f), d).split(os.sep)[0], name[:70])
            counters[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur = per_dispatch.get((f.replace("counter_collection", "X"), r.get("Dispatch_Id")))
                if dur:
                    counters[key]["effective_clock_GHz"].append(float(r["Counter_Value"]) / (dur * 1e3))
    return counters, durs

for d in sys.argv[1:]:
    counters, durs = load(d)
    print("#", d)
    for key in sorted(set(counters) | set(durs)):
        print("==", key)
        if key in durs:
            v = durs[key]
            print(f"   {'duration_us':40s} avg {sum(v)/len(v):16.2f}  n={len(v)}  min {min(v):.2f}")
        for c, v in sorted(counters.get(key, {}).items()):
            print(f"   {c:40s} avg {sum(v)/len(v):16.3f}  n={len(v)}")
