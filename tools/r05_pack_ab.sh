# same-box A/B of the packed kernel's K = 10 instances compiled for 3 (product) and 4 (variant _pw4) wavefronts per SIMD
set -u
for v in "" "_pw4"; do
  echo "== library variant '${v}'"
  EEA_LIB_VARIANT=$v python tools/pack_sweep.py --spl 50 --batches ${BATCHES:-12288,16384,24576,32768} --lanes ${LANES:-16,8} 2>&1 | grep -v "configs\[0\]\|automatic"
done
