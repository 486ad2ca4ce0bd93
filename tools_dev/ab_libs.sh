# A/B of two builds of the library (libssd_hip_base.so / libssd_hip_new.so next to libssd_hip.so) on the given ab_conv_multi specs
for rep in 1 2; do
for lib in libssd_hip_base.so libssd_hip_new.so; do
  echo "== $lib"
  AB_LIB=$lib python tools_dev/ab_conv_multi.py "$@" 2>/dev/null
done
done
