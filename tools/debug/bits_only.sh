export TMPDIR=/tmp
mkdir -p gpurun_out
( SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_oldsteps.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_old.npz
  SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_wetasm0.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_wet0.npz
  SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_old.so timeout 900 python tools/debug/steps_bits.py dump /tmp/b_modes0.npz
  timeout 900 python tools/debug/steps_bits.py dump /tmp/b_new.npz
  echo "== -DSMART_STEP_ARMS=0 (compiled step loop of round 2) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_old.npz /tmp/b_new.npz
  echo "== -DSMART_WET_ASM=0 (hipcc's wet-interval loop) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_wet0.npz /tmp/b_new.npz
  echo "== -DSMART_WET_MODES=0 -DSMART_RAIN_FILL_EXIT=0 (every wet step fills all six layers) against the shipped library"
  python tools/debug/steps_bits.py compare /tmp/b_modes0.npz /tmp/b_new.npz ) > gpurun_out/r03_steps_bits.txt 2>&1
grep -c . gpurun_out/r03_steps_bits.txt; grep "differ\|==" gpurun_out/r03_steps_bits.txt
