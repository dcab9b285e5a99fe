cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in 512 768 1024; do for MX in 256 4096; do
rm -rf /tmp/pp; IFA_PREFILL_MID_MAX=$MX timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o st -- python $R/tools/prefill_steps.py llama2_7b $T 4 > /tmp/pp.log 2>&1
f=$(find /tmp/pp -name "*kernel_stats.csv" | head -1)
echo "== T=$T mid_max=$MX: $(grep 'prefill ' /tmp/pp.log | tail -1)"; python3 -c "
import csv,sys
for r in list(csv.reader(open('$f')))[1:]:
    if 'k_gemm_mid' in r[0] or 'k_gemm_big' in r[0]: print('   %-60s calls %4s  avg %8.1f us' % (r[0].replace('(ifa::GmArgs, ifa::BigGeo)','').replace('void ifa::',''), r[1], float(r[3])/1000))
"
done; done
