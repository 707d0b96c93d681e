#!/bin/bash
# compile statmc_filter_sym.hip to ISA and print resource usage + instruction histogram of the hot loops
cd /root/repo/statmc_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-misched=0 -S --cuda-device-only ${1:-statmc_filter_sym.hip} -o /tmp/sym.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A9 "window_filter_sym" | grep -E "VGPRs:|Scratch|error"
awk '/^_ZN6statmc3sym17window_filter_symILb1EEEvNS_10FilterArgsE:/,/s_endpgm/' /tmp/sym.s > /tmp/sym_k.s
for L in $(grep "Parent Loop" /tmp/sym_k.s | awk '{print $1}' | tr -d ':'); do echo "== $L"; sed -n "/^${L}:/,/s_cbranch/p" /tmp/sym_k.s | grep -v "^\s*;" | awk '{print $1}' | sort | uniq -c | sort -rn | head -18 | tr '\n' ' '; echo; done
echo "op_sel: $(grep -c op_sel /tmp/sym_k.s)  scratch: $(grep -c scratch_ /tmp/sym_k.s)"
