# On the GPU box: failure rate of the cluster chain's float-converter instance (tuning builds, tools/variant.sh <tag> f8_cchain -DF8_CC_FLOAT_INSTANCE ...) in the TAIL soak
for lib in "$@"; do
  echo "== $lib: $(F8NET_LIB=f8net_amd/libf8net_$lib.so timeout 300 python tools/soak_chain7.py 150 2>&1 | grep "requant_float=1" | sed -E 's/TAIL form, //' | tr '\n' ';')"
done
