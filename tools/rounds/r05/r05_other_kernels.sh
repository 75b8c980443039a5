cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/po; MF_ONLY=step timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/po -- python3 $GRAFT_REPO_ROOT/tools/time_moco_step.py 1024 > /tmp/po.log 2>&1
python3 - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("/tmp/po/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
steps=None
for r in rows:
    if "render_kernel" in r["Name"]: steps=int(r["Calls"])/2
print("steps", steps, "total ms/step", tot/1e6/steps)
for r in sorted(rows, key=lambda r:-float(r["TotalDurationNs"])):
    if not r["Name"].startswith(("mf::","void mf::")):
        print(f'{float(r["TotalDurationNs"])/1e3/steps:8.1f} us/step  {int(r["Calls"])/steps:6.1f} calls/step  avg {float(r["AverageNs"])/1e3:7.1f}  {r["Name"][:150]}')
PY
