# role / stage timings of the small-launch kernels: a -DTMX_TINY_PROF build (bash tools/build_variant.sh tinyprof -DTMX_TINY_PROF) with roles
# switched off by TMX_TINY_DBG; k_eddsa = k_tiny, k_verdict = k_tiny_tail (HIP events on the dispatches)
cd $GRAFT_REPO_ROOT
export TMX_LIB=build_ab/tinyprof.so QUICK=1
for d in 0 0x1e 0x11e 0x21e 0x41e 0x1d 0x1b 0x17 0x0f 0x1f 0x1000 0x2000 0x4000 0x8000 0xc000; do
  TMX_TINY_DBG=$d timeout 120 python tools/tiny_probe.py 2>&1 | grep DBG
done
