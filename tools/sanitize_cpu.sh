#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the CPU-side code (GPU sanitizers are not
# available on the pool): the host mirror's own test driver (reference KATs on the C++ classes), the
# map_server loader on a map file given as $1 (optional), and the C oracle through one closed-loop case.
set -e
set -o pipefail
export UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/eea_asan
mkdir -p "$OUT"
FLAGS="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer"
HOST=$ROOT/ergodic_exploration_amd/host
LIB=$ROOT/ergodic_exploration_amd/lib
LINK="-L$LIB -lergodic_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$LIB -Wl,-rpath,/opt/rocm/lib"
INC="-std=c++17 -D__HIP_PLATFORM_AMD__ -I$HOST/include -I$ROOT/include -I/opt/rocm/include"
g++ $FLAGS $INC $HOST/test/host_tests.cpp -o $OUT/host_tests $LINK
ASAN_OPTIONS=detect_leaks=0 $OUT/host_tests cpu | tail -1
if [ -n "$1" ]; then
  g++ $FLAGS $INC $HOST/src/exploration_omni_node.cpp -o $OUT/exploration_omni $LINK
  ASAN_OPTIONS=detect_leaks=0 $OUT/exploration_omni --map-yaml "$1" --dump-map | head -1
fi
cat > $OUT/driver.c <<'C'
#include "ergodic_oracle.h"
#include <stdio.h>
int main(void) {
  eo_control_config cfg = { EO_MODEL_OMNI, 0.1, 5.0, 0.1, 1.0, 10, {1,0,0,0,1,0,0,0,2}, {-1,-1,-2}, {1,1,2} };
  double mu[4] = {2.5,2.5,8.5,2.5}, sg[4] = {1.5,1.5,1.5,1.5};
  eo_control* ec; if (eo_control_create(&cfg, &ec)) return 1;
  eo_control_set_target(ec, 2, mu, sg);
  double x[3] = {1,1,0.3}, u[3], mem[21];
  for (int i = 0; i < 21; ++i) mem[i] = 1.0 + 0.01 * i;
  for (int c = 0; c < 3; ++c) { if (eo_control_step(ec, -1, 11, -1, 5, x, mem, 7, u, 0)) return 2; }
  double poses[12] = {1,1,0.3, 2,2,0.1, 3,1,0.2, 4,2,-0.5}, ul[12];
  eo_bench_control(&cfg, 2, mu, sg, -1, 11, -1, 5, poses, 4, 2, 2, ul);
  printf("oracle under ASAN/UBSAN: u = %.17g %.17g %.17g\n", u[0], u[1], u[2]);
  eo_control_destroy(ec); return 0;
}
C
gcc $FLAGS -ffp-contract=off -I$ROOT/oracle $OUT/driver.c $ROOT/oracle/ergodic_oracle.c -o $OUT/oracle_driver -lm -lpthread
$OUT/oracle_driver
echo "sanitizers: clean"
