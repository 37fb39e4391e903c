#!/bin/bash
# columns per lane (EEA_PHIK_CPT) sweep of the phi_k streaming kernel; "0" = the built-in policy
for c in 1 2 4 0; do
  echo "== EEA_PHIK_CPT=$c"
  if [ "$c" = "0" ]; then unset EEA_PHIK_CPT; else export EEA_PHIK_CPT=$c; fi
  bash tools/phik_prof.sh
done
