export PYTHONDONTWRITEBYTECODE=1
bash tools/prof.sh r02g > gpurun_out/prof_r02g.log 2>&1
tail -3 gpurun_out/prof_r02g.log
python - <<'PY'
import csv, glob, json
f = glob.glob("gpurun_out/prof_r02g/kt/**/*kernel_stats.csv", recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print("%-90s %5s calls avg %9.1f us %5.1f%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
s = json.load(open("gpurun_out/prof_r02g/pmc_summary.json"))
for k, d in s.items():
    if "nuts_gaussian" in k or "glm_planes" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-30s %16.1f" % (c, v["mean"]))
print(open("gpurun_out/prof_r02g/traffic.json").read() if glob.glob("gpurun_out/prof_r02g/traffic.json") else "no traffic.json")
PY
