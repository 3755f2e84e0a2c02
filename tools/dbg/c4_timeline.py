"""per-kernel timeline of single K9 projections inside a config-4 solve, from a rocprofv3 kernel-trace database (rocpd sqlite):
    python tools/dbg/c4_timeline.py <run_results.db> [projection indices ...]"""
import sqlite3, sys
import numpy as np
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0] for r in rows]
dur = np.array([(r[2] - r[1]) / 1e3 for r in rows])
fronts = [i for i, nm in enumerate(names) if "k_psd_front" in nm]
r2 = [i for i, nm in enumerate(names) if "k_psd_gemm<3>" in nm]
per = np.array([(rows[b][2] - rows[a][1]) / 1e3 for a, b in zip(fronts, r2)])
print("projections %d; wall per projection (front start -> R2 end), by iteration range:" % len(per))
for lo, hi in ((5, 100), (105, 225), (260, 400), (400, 640)):
    if len(per) > lo:
        print("   [%d, %d): mean %.0f us" % (lo, min(hi, len(per)), per[lo:hi].mean()))
sw = np.array([dur[i] for i, nm in enumerate(names) if "k_psd_sweep_mc" in nm])
print("k_psd_sweep_mc launches %d; histogram of durations (us) [0,5,15,30,60,120,400,800,1300,2000]:" % len(sw), np.histogram(sw, bins=[0, 5, 15, 30, 60, 120, 400, 800, 1300, 2000])[0])
for it in [int(x) for x in sys.argv[2:]]:
    if it >= len(fronts):
        continue
    a, b = fronts[it], r2[it]
    print("projection %d: wall %.0f us" % (it, (rows[b][2] - rows[a][1]) / 1e3))
    print("   " + "  ".join("%s %.1f" % (names[i].split("(")[0].replace("scship::", "").replace("void ", "").replace("k_psd_", "").replace("k_proj_", "")[:12], dur[i]) for i in range(a, b + 1)))
