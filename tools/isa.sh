#!/bin/bash
# Dumps the gfx950 ISA of one kernel of a csrc/*.hip file:  tools/isa.sh gru.hip _Z10gru_bwd_b3ILi128EEv7GruArgsi > out.s
set -e
SRC=/root/repo/multimodalsignal_amd/csrc/$1
OUT=/tmp/isa_$(basename $1 .hip).s
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -S --cuda-device-only $SRC -o $OUT 2>/dev/null
awk -v k="^$2:" '$0 ~ k {p=1} p {print} p && /s_endpgm/ {exit}' $OUT
