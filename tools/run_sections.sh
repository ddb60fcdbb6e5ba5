for x in 3_0 0_4 4_5 5_6 6_7 12_1 1_2; do
  PG_STAMPS=1 PG_ALIGN_RING=1 PAGAN_DP_LIB=$PWD/pagan2-msa_amd/libpagan_dp_sec_$x.so timeout -k 10 200 python tests/diagnostics/probe_pair.py 32 > gpurun_out/r03_sec_$x.log 2>&1
  echo "== $x: $(grep '^node' gpurun_out/r03_sec_$x.log | tail -1 | cut -c1-60)"
  grep "^assist 0: [0-9]* diagonals" gpurun_out/r03_sec_$x.log | sed 's/prepare (descriptors, loader) [0-9]*, //' | cut -c1-110
done
