# sweeps of clx_ms_assign_dense: points per lane (CLX_MS_DENSE_U) and, with libraries built with -DCLX_DENSE_WAVES=2/4
# next to the default one, wave-tiles per block
for lib in libclx.so libclx_w2.so libclx_w4.so; do
  [ -f cellulus_amd/$lib ] || continue
  for sz in 4096 8192; do echo $lib $sz; CLX_LIB=cellulus_amd/$lib python tools/bench_stream.py $sz 2>/dev/null | grep "ms_assign " | cut -c1-70; done
done
