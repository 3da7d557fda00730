#!/bin/bash
# timeline of ONE hipGraph replay of the train_cap step (two compute streams): rocprofv3 --kernel-trace over the captured bench, the last
# replay's dispatches binned in 100 us windows -> gpurun_out/<tag>_timeline.txt (busy fraction, kernels in flight, workgroups in flight)
TAG=${1:-x}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$TAG -o step -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-clock-probe > $R/gpurun_out/${TAG}_tl_run.log 2>&1; echo rc=$?
cd $R
f=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" | head -1)
python - "$f" gpurun_out/${TAG}_replay_dispatches.csv > gpurun_out/${TAG}_timeline.txt <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "rng_advance" in r["Kernel_Name"]]
# the last replay = everything that starts after the previous step's Adam has finished (branches of the captured step that fork from its
# beginning -- the weight-plane refresh, descriptor tables -- may start before its first kernel on the main stream does)
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] and idx and i < idx[-1]]
if idx and ad:
    cut = int(rows[ad[-1]]["End_Timestamp"])
    last = [r for r in rows[ad[-1] + 1:] if int(r["Start_Timestamp"]) >= cut]
else:
    last = rows[idx[-1]:] if idx else rows
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
ev = []
for r in last:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    g = lambda k: int(r.get(k, 1) or 1)
    wg = (g("Grid_Size_X") * g("Grid_Size_Y") * g("Grid_Size_Z")) // max(1, g("Workgroup_Size_X") * g("Workgroup_Size_Y") * g("Workgroup_Size_Z"))
    ev.append((s, e, wg, r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0][:40]))
span = (t1 - t0) / 1e3
print(f"# last replay: {len(ev)} dispatches, span {span / 1e3:.3f} ms, sum of kernel durations {sum(e - s for s, e, _, _ in ev) / 1e6:.3f} ms")
BIN = 100_000
nb = (t1 - t0 + BIN - 1) // BIN
busy = [0] * nb; act = [0] * nb; wgs = [0] * nb; names = [collections.Counter() for _ in range(nb)]
# union busy via sweep
pts = sorted([(s, 1) for s, e, _, _ in ev] + [(e, -1) for s, e, _, _ in ev])
cur = 0; prev = 0; union = 0
for t, d in pts:
    if cur > 0:
        a, b = prev, t
        while a < b:
            k = a // BIN; nxt = min(b, (k + 1) * BIN)
            busy[k] += nxt - a; a = nxt
        union += t - prev
    cur += d; prev = t
for s, e, wg, n in ev:
    a = s
    while a < e:
        k = a // BIN; nxt = min(e, (k + 1) * BIN)
        act[k] += nxt - a; wgs[k] += (nxt - a) * min(wg, 512); names[k][n] += nxt - a; a = nxt
print(f"# GPU busy (>= 1 kernel) {union / 1e6:.3f} ms of {span / 1e3:.3f}")
# every kernel of the replay by name: what a captured step consists of (no framework kernel is among them)
cnt = collections.Counter(); dur = collections.Counter()
for s_, e_, _, n_ in ev:
    cnt[n_] += 1; dur[n_] += e_ - s_
fw = sum(c for n_, c in cnt.items() if n_.startswith(("at::", "void at::", "__amd_rocclr")))
print(f"# kernels from the replay's first launch to the end of the trace: {len(cnt)} distinct names, {fw} launches of framework / runtime kernels "
      f"(at::*, __amd_rocclr_*; the bench reads the loss back after the last replay: 2 copies that are not graph nodes)")
for n_, c in sorted(cnt.items(), key=lambda kv: -dur[kv[0]]):
    print(f"#   {c:4d} x {n_:42s} {dur[n_] / 1e3:9.1f} us")
# every dispatch of the replay: start / end relative to the first launch, queue, workgroups (the critical path is read off this list)
with open(sys.argv[2], "w") as fo:
    fo.write("start_us,end_us,dur_us,queue,workgroups,kernel\n")
    q = {}
    for r in last:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        g = lambda k: int(r.get(k, 1) or 1)
        wg = (g("Grid_Size_X") * g("Grid_Size_Y") * g("Grid_Size_Z")) // max(1, g("Workgroup_Size_X") * g("Workgroup_Size_Y") * g("Workgroup_Size_Z"))
        qi = q.setdefault(r.get("Queue_Id", "0"), len(q))
        n = r["Kernel_Name"].replace("void ", "", 1).replace("(anonymous namespace)::", "").split("(")[0][:60]
        fo.write(f"{s / 1e3:.2f},{e / 1e3:.2f},{(e - s) / 1e3:.2f},{qi},{wg},\"{n}\"\n")
print("t_ms  busy  kernels_in_flight  workgroups_in_flight(capped 512/kernel)  top kernels")
for k in range(nb):
    top = ", ".join(f"{n}:{v / BIN:.2f}" for n, v in names[k].most_common(3))
    print(f"{k * BIN / 1e6:5.1f}  {busy[k] / BIN:4.2f}  {act[k] / BIN:5.2f}  {wgs[k] / BIN:7.0f}   {top}")
PY
head -3 gpurun_out/${TAG}_timeline.txt
rm -rf /tmp/tl_$TAG
