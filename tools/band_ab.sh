# NOTE (round 5): needs a -DSYN3R_TUNING build of the library (tools/build_variant.sh tune -DSYN3R_TUNING; SYN3R_LIB_OVERRIDE=abtmp/libtune.so)
for b in 1 2 3 4; do echo "SYN3R_Z_BAND=$b"; SYN3R_Z_BAND=$b SYN3R_GEMM_Z=1 python tools/gemm_iso3.py - 2>&1 | grep -v amdgpu.ids | cut -c26-420; done
