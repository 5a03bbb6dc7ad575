"""Per-step launch timeline from a rocprofv3 --kernel-trace CSV: start / end of every kernel relative to the step's
first launch, for the last N steps (a step starts at each ode_step_kernel that follows an update kernel)."""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
name = lambda r: re.sub(r"\(.*", "", re.sub(r"^void |\(anonymous namespace\)::", "", r["Kernel_Name"]))[:44]
starts = []
prev = ""
for i, r in enumerate(rows):
    nm = name(r)
    if nm.startswith("ode_step") and not prev.startswith("ode_step") and not prev.startswith("emi_rows"):
        starts.append(i)
    if nm.startswith("emi_rows") and not prev.startswith("ode_step") and not prev.startswith("emi_rows"):
        starts.append(i)
    prev = nm
starts = sorted(set(starts))
for a, b in list(zip(starts[:-1], starts[1:]))[-n_steps:]:
    t0 = int(rows[a]["Start_Timestamp"])
    print(f"--- step: {b - a} launches, {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us to the next step")
    for r in rows[a:b]:
        print(f"  {name(r):46s} stream {r.get('Stream_Id', '?'):>3s} queue {r.get('Queue_Id', '?'):>3s}  "
              f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} -> {(int(r['End_Timestamp']) - t0) / 1e3:8.1f} us")
