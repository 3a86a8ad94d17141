#!/bin/bash
# tools/check_spills.sh <obj-dir> <expectations-file> [--print-all]
#
# Guards the register allocation of the hot kernels.  The frame loops of the GRU kernels are written against a register
# budget (512 VGPR + AGPR per lane, weights parked in AGPRs for the whole launch): one more live value and hipcc spills
# into scratch inside the loop -- still correct, silently slower, and nothing else in the build would say so.  This reads
# the code-object metadata (.vgpr_spill_count, .sgpr_spill_count, .private_segment_fixed_size) of every kernel in
# <obj-dir>/*.o and compares the instantiations listed in <expectations-file> with what is committed there:
#
#     <demangled kernel name, exact>|<vgpr spills, exact>|<max sgpr spills>|<max scratch bytes>
#
# A VGPR spill count that differs from the expectation -- or an SGPR spill count / scratch size above it -- fails the build
# (exit 1); so does a listed kernel that no longer exists.  Run by keyword_spotting_amd/csrc/Makefile after every link;
# `make SPILL_CHECK=` skips it (tools/build_variant.sh does: instrumented variants allocate differently on purpose).
set -e
OBJ=$1; EXP=$2; ALL=$3
LLVM=${LLVM_BIN:-/opt/rocm/lib/llvm/bin}
T=$(mktemp -d); trap 'rm -rf $T' EXIT
: > $T/all.txt
for o in $OBJ/*.o; do
  b=$(basename $o .o)
  $LLVM/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/$b.fat 2>/dev/null || continue
  [ -s $T/$b.fat ] || continue
  tgt=$($LLVM/clang-offload-bundler --list --input=$T/$b.fat --type=o | grep amdgcn | head -1)
  [ -n "$tgt" ] || continue
  $LLVM/clang-offload-bundler --unbundle --input=$T/$b.fat --type=o --targets=$tgt --output=$T/$b.co
  # one line per kernel: mangled|vgprs|agprs|vgpr spills|sgpr spills|scratch
  $LLVM/llvm-readelf --notes $T/$b.co | awk '
    /\.agpr_count:/ {a=$NF} /\.name:/ {n=$NF} /\.private_segment_fixed_size:/ {p=$NF} /\.sgpr_spill_count:/ {s=$NF}
    /\.vgpr_count:/ {v=$NF} /\.vgpr_spill_count:/ {print n "|" v "|" a "|" $NF "|" s "|" p}' >> $T/mangled.txt
done
cut -d'|' -f1 $T/mangled.txt | c++filt | sed -e 's/^void //' -e 's/(.*$//' > $T/names.txt
paste -d'|' $T/names.txt <(cut -d'|' -f2- $T/mangled.txt) | sort -u > $T/all.txt
if [ "$ALL" == "--print-all" ]; then
  echo "kernel|vgprs|agprs|vgpr_spills|sgpr_spills|scratch_bytes"; cat $T/all.txt; exit 0
fi
fail=0
printf '%-62s %5s %5s %11s %11s %8s\n' "register allocation of the hot kernels" vgpr agpr "vgpr-spill" "sgpr-spill" scratch
while IFS='|' read -r name want_v max_s max_p; do
  case "$name" in ''|\#*) continue;; esac
  line=$(grep -F "$name|" $T/all.txt | head -1 || true)
  if [ -z "$line" ]; then echo "check_spills: kernel '$name' is listed in $EXP but not in the build"; fail=1; continue; fi
  IFS='|' read -r _n v a sv ss sp <<< "$line"
  verdict=ok
  if [ "$sv" != "$want_v" ]; then verdict="VGPR SPILLS $sv != expected $want_v"; fail=1; fi
  if [ "$ss" -gt "$max_s" ]; then verdict="$verdict; SGPR spills $ss > $max_s"; fail=1; fi
  if [ "$sp" -gt "$max_p" ]; then verdict="$verdict; scratch $sp B > $max_p"; fail=1; fi
  printf '%-62s %5s %5s %11s %11s %8s  %s\n' "$name" "$v" "$a" "$sv" "$ss" "$sp" "$verdict"
done < $EXP
if [ $fail -ne 0 ]; then
  echo "check_spills: the register allocation of a hot kernel changed (see above).  If the change is intended, update $EXP in the same commit."
  exit 1
fi
