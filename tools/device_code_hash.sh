#!/bin/bash
# sha256 of the gfx950 instruction stream of every device translation unit of libmplx.so (llvm-objdump -d of the code object inside the
# .o's fat binary): two builds whose hashes agree run the same kernels.  usage: tools/device_code_hash.sh [dir with the .o files]
B=/opt/rocm/lib/llvm/bin
D=${1:-$(dirname "$0")/../mpl_ros_amd/csrc}
T=$(mktemp -d)
for o in mplx_help_launch mplx_spec_launch mplx_filter_launch mplx_yaw_launch mplx_lpa_launch mplx_poly_launch mplx_poly_lpa mplx_api; do
  [ -f $D/$o.o ] || continue
  $B/llvm-objcopy -O binary --only-section=.hip_fatbin $D/$o.o $T/$o.fatbin
  $B/clang-offload-bundler --unbundle --type=o --input=$T/$o.fatbin --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$o.co 2>/dev/null
  # (instruction lines only, without the address / encoding comment: symbol names -- a template that gained a defaulted parameter --
  #  and the file name do not count)
  h=$($B/llvm-objdump -d $T/$o.co | grep -P "^\t" | sed 's#[ \t]*//.*$##' | sha256sum | cut -c1-16)
  n=$($B/llvm-objdump -d $T/$o.co | grep -cP "^\t")
  echo -n "[$n instructions] "
  echo "$o $h"
done
rm -rf $T
