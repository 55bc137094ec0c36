for rep in 1 2 3; do
  for so in default tools/variants/libsmart_amd_w3.so; do
    if [ "$so" = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/$so; fi
    echo "== $so"; python tools/debug/flat_only.py 100000 4 2>/dev/null | tail -5 | tr '\n' ' '; echo
    python tools/debug/flat_only.py 1000000 3 objectives 2>/dev/null | tail -4 | tr '\n' ' '; echo
    python tools/debug/flat_only.py 300000 3 objectives 2>/dev/null | tail -4 | tr '\n' ' '; echo
  done
done
