#!/bin/bash
# Timing experiments on the hand-scheduled loop: builds pagan2-msa_amd/libpagan_dp_exp_<x>.so for every PG_HOT_EXP
# variant named on the command line (tools/gen_hot_asm.py: the variants drop parts of the step -- WRONG RESULTS, same
# schedule), then regenerates the product's loop.  Use with PAGAN_DP_LIB=... tests/diagnostics/probe_pair.py.
set -e
cd "$(dirname "$0")/.."
# whatever happens below, csrc/ ends up with the PRODUCT's loop again (a variant that fails to build must not stay there)
trap 'python tools/gen_hot_asm.py' EXIT
for x in "$@"; do
    PG_HOT_EXP=$x python tools/gen_hot_asm.py
    python - "$x" <<'PY'
import importlib.util, sys
spec = importlib.util.spec_from_file_location("_b", "pagan2-msa_amd/build.py"); m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
import os
m.FLAGS += os.environ.get("PG_EXP_DEFS", "").split()     # e.g. PG_EXP_DEFS=-DPG_ASSIST_IDLE (assist waves return at once: only with variants that wait for none)
print(m.build(force=True, out=m.HERE + "/libpagan_dp_exp_%s%s.so" % (sys.argv[1], os.environ.get("PG_EXP_TAG", ""))))
PY
done
