LINSYS=hip_dense MAXIT=5 SCS_HIP_POOL_MB=16384 SCS_HIP_DEBUG=setup timeout 300 python tools/batch_leg.py 512 16 1 > /tmp/il.txt 2>&1
python3 - <<'PY'
import re,collections
acc=collections.defaultdict(float); n=collections.Counter()
for l in open('/tmp/il.txt'):
    m=re.match(r"\[scs-hip setup\] (.*?)\s+([0-9.]+) ms",l)
    if m: acc[m.group(1)]+=float(m.group(2)); n[m.group(1)]+=1
for k,v in acc.items(): print("%-60s sum %8.1f ms over %d members (%.2f ms each)"%(k,v,n[k],v/n[k]))
print(open('/tmp/il.txt').read().strip().splitlines()[-1][:400])
PY
