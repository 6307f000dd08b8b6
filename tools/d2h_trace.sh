#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r03/tr
mkdir -p $O
for cfg in "256 2 before" "256 1 before"; do
  n=$(echo $cfg | tr ' ' '_')
  rocprofv3 --kernel-trace --memory-copy-trace -d $O/$n -o run --output-format csv -- python3 tools/d2h_trace.py $cfg > $O/log_$n.txt 2>&1
  grep fps $O/log_$n.txt
done
python3 - <<'PY'
import csv, glob
for n in ("256_2_before", "256_1_before"):
    ev = []
    for f in glob.glob("gpurun_out/r03/tr/%s/**/*kernel_trace.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:14], r.get("Queue_Id", "?")))
    for f in glob.glob("gpurun_out/r03/tr/%s/**/*memory_copy_trace.csv" % n, recursive=True):
        for r in csv.DictReader(open(f)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")[-14:], "-"))
    ev.sort()
    t0 = ev[0][0]
    big = [e for e in ev if e[2].startswith("COPY") and e[1] - e[0] > 100000]
    print(n, "events", len(ev), "big copies", len(big))
    # the last 5 big copies and the kernels around them
    if len(big) >= 6:
        lo, hi = big[-6][0], big[-2][1]
        for e in ev:
            if lo - 700000 <= e[0] <= hi:
                print("   %9.1f %9.1f  %-20s q=%s" % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, e[2], e[3]))
PY
