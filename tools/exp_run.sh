for x in "$@"; do
  echo "== lib$x"
  PG_ALIGN_RING=1 PAGAN_DP_LIB=$PWD/pagan2-msa_amd/libpagan_dp$x.so timeout -k 10 120 python tests/diagnostics/probe_pair.py ${LEAVES:-2} 2>&1 | grep "^node" | tail -1
done
