mkdir -p gpurun_out/ab
python tools/et_probe.py run --B 16 --N 256 > gpurun_out/ab/et_probe_roll.txt 2>&1; python - <<'PY'
import re
tot=[0]*4; rows=[]
for l in open('gpurun_out/ab/et_probe_roll.txt'):
    m=re.match(r"slot\s+(\d+) stage.*?:\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)\s+(-?\d+)", l)
    if m and int(m.group(1))<239:
        v=[int(m.group(k)) for k in range(2,6)]; rows.append((int(m.group(1)), v))
        for k in range(4): tot[k]+=v[k]
print("sum over slots 0..238:", tot)
print("slots 186..230:", [(r[0], max(r[1])) for r in rows if 186<=r[0]<=230])
PY
