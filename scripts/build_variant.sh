#!/bin/bash
# A/B builds of the library: scripts/build_variant.sh <name> [extra hipcc flags...] -> scripts/bin/libtfQMRgpu_<name>.so
# (all objects but tfq_spmm.o are taken from the regular build -- from its lab objects when -DTFQ_LAB is among the flags; TFQMRGPU_LIB=<path> makes the Python binding load a variant;
#  SPMM_SRC=<file> compiles another source file in place of tfq_spmm.hip, e.g. an earlier revision from `git show`)
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${SPMM_SRC:-$root/tfqmrgpu_amd/csrc/tfq_spmm.hip}
# timing-only variants (-DTFQ_PROBE=..., -DTFQ_LAB_CLOCK, -DTFQ_LAB_STAMPS, -DTFQ_B8_DEPTH=...) exist only in the lab copy of the multiply (r04: the product source carries none of them)
case " $* " in *TFQ_PROBE*|*TFQ_LAB_CLOCK*|*TFQ_LAB_STAMPS*|*TFQ_B8_DEPTH*) src=${SPMM_SRC:-$root/scripts/lab/tfq_spmm_probes.hip};; esac
# a variant with -DTFQ_LAB is linked against the LAB objects of the other files: one definition of tfq::lab_switch per library
objdir=$root/tfqmrgpu_amd/lib/obj
case " $* " in *"-DTFQ_LAB "*|*"-DTFQ_LAB") objdir=$root/tfqmrgpu_amd/lib/obj/lab;; esac
mkdir -p $root/scripts/bin/obj_$name
make -s -C $root/tfqmrgpu_amd/csrc >/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -I$root/include -I$root/tfqmrgpu_amd/csrc -Wall -Wno-unused-function --offload-arch=gfx950 \
   -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -x hip -c $src -o $root/scripts/bin/obj_$name/tfq_spmm.o
objs=""
for o in tfq_api tfq_vec tfq_layout tfq_plan tfq_shard tfq_error tfq_order; do objs="$objs $objdir/$o.o"; done
objs="$objs $root/tfqmrgpu_amd/lib/obj/tfq_fortran.o"
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $root/scripts/bin/libtfQMRgpu_$name.so $root/scripts/bin/obj_$name/tfq_spmm.o $objs -Wl,-soname,libtfQMRgpu.so.1 -ldl
rm -rf $root/scripts/bin/obj_$name
echo built scripts/bin/libtfQMRgpu_$name.so
