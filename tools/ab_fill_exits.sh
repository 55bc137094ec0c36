export TMPDIR=/tmp
for rep in 1 2; do for so in default f2 f3 f4; do
if [ $so = default ]; then unset SMART_AMD_LIB; else export SMART_AMD_LIB=$PWD/smartpy_amd/csrc/libsmart_amd_$so.so; fi
echo "== $so"; python tools/debug/sort_rows.py 1000000 200000 2>&1 | grep "T in 64 bins, then S\*Z " | awk 'NR%2==0'
done; done
