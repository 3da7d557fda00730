#!/bin/bash
# ISA census of one kernel without a GPU: compiles <file> (an experiment .hip meant for tools/experiments/exp_lib.hip's translation unit, or
# /dev/null for a kernel of the product's own attention_bf16.hip) behind the product's attention_bf16.hip, prints registers / scratch and the
# instruction mix of its MFMA loops.
#   usage: tools/exp_isa.sh tools/experiments/attn_fwd32.hip attn_fwd32 [extra -D flags]      tools/exp_isa.sh /dev/null dkvg8_kernelILi256
F=$(realpath "$1"); K=$2; shift 2
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/expisa.XXXX)
cat > $T/tu.hip <<EOT
#define BMT_EXP_LIB 1
#include "$R/bmt_amd/csrc/attention_bf16.hip"
#include "$F"
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc --cuda-device-only -I$R/bmt_amd/csrc "$@" -Rpass-analysis=kernel-resource-usage -S $T/tu.hip -o $T/tu.s 2> $T/res.txt || { grep -v "remark:\|argument unused" $T/res.txt | head -30; exit 1; }
python - "$T/res.txt" "$K" <<'EOP'
import re,sys
b=open(sys.argv[1]).read()
for blk in re.split(r"remark: Function Name: ", b)[1:]:
    m=blk.split()[0]
    if sys.argv[2] in m:
        g=lambda k:(re.search(re.escape(k)+r": (\d+)",blk) or [None,"?"])[1]
        print(m[:70], 'VGPR',g('VGPRs'),'AGPR',g('AGPRs'),'scratch',g('ScratchSize [bytes/lane]'),'SGPR',g('SGPRs'))
EOP
python $R/tools/isa_loops.py $T/tu.s "$K" 8
echo "listing: $T/tu.s"
