#!/bin/bash
# gemm_nt256d_kernel timing variants: every argument is "name" or "name:-DFLAG=.."; a bare number is OD_ABLATE bits (gemm_nt_owd.hip;
# results wrong on purpose).  Builds scratch/pw_abl/libmaest_<name>.so locally; scratch/owd_ablate_run.py times them on the GPU box.
set -e
cd "$(dirname "$0")/.."
python maest_amd/build.py >/dev/null
rm -rf scratch/pw_abl; mkdir -p scratch/pw_abl
for a in "$@"; do
  name=${a%%:*}; flags=""; [[ "$a" == *:* ]] && flags=${a#*:}
  [[ "$name" =~ ^[0-9]+$ ]] && flags="$flags -DOD_ABLATE=$name"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result $flags \
      -c maest_amd/csrc/gemm_nt_owd.hip -o scratch/pw_abl/owd_$name.o &
done
wait
for a in "$@"; do
  name=${a%%:*}
  objs=$(ls maest_amd/build/*.hip.o | grep -v gemm_nt_owd)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/pw_abl/libmaest_$name.so $objs scratch/pw_abl/owd_$name.o
  rm scratch/pw_abl/owd_$name.o
done
ls scratch/pw_abl | tr '\n' ' '
