# deferred-store ring kernel (pd): exact layout + epilogues, then the A/B table
O=gpurun_out/r04l; mkdir -p $O
MVPTR_LIB=diag NTP_CFG=pd timeout 600 python3 tools/debug_ntp.py > $O/debug_pd.txt 2>&1; grep -E "differ|cfg|Error|error" $O/debug_pd.txt | head -20
MVPTR_LIB=diag NTP_CFG=p timeout 600 python3 tools/debug_ntp.py > $O/debug_p.txt 2>&1; grep -E "differ|cfg|Error|error" $O/debug_p.txt | head -20
MVPTR_LIB=diag timeout 600 python3 tools/blas_table.py --ab --cfg pd --ms 10917,37748,64000 > $O/blas_pd.txt 2>&1; cat $O/blas_pd.txt
