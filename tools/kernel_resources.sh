#!/bin/bash
# VGPRs / scratch bytes per lane of every kernel of the convolution family (and of any other unit given), from the compiler's
# -Rpass-analysis=kernel-resource-usage remarks.   bash tools/kernel_resources.sh [unit.hip ...] [-- extra hipcc flags]
# Prints "scratch vgprs kernel", kernels with scratch first.  (Round 5: the per-case store loops pushed several kernels into scratch.)
cd "$(dirname "$0")/../pasta-gan_amd/csrc"
UNITS=(); FLAGS=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; FLAGS=("$@"); break; fi; UNITS+=("$1"); shift; done
[ ${#UNITS[@]} -eq 0 ] && UNITS=(conv_tu_*.hip pieces.hip)
TMP=$(mktemp -d)
for f in "${UNITS[@]}"; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -I ../../include "${FLAGS[@]}" -c "$f" -o "$TMP/$f.o" -Rpass-analysis=kernel-resource-usage 2>&1 \
      | grep -E "Function Name|  VGPRs:|ScratchSize" | paste - - - \
      | sed -E 's/.*Name: ([^ ]*) .*VGPRs: ([0-9]+) .*ScratchSize \[bytes\/lane\]: ([0-9]+).*/\3 \2 \1/' > "$TMP/$f.txt" ) &
done
wait
cat "$TMP"/*.txt | sort -rn | while read s v n; do echo "$s $v $(echo "$n" | c++filt | sed 's/pasta:://g')"; done
rm -rf "$TMP"
