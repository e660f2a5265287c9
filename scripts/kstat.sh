#!/bin/bash
# scripts/kstat.sh <file.s> <mangled-kernel-prefix>: registers, scratch, spill traffic and instruction counts of one kernel
f=$1; k=$2
awk -v k="$k" 'index($0,k)==1 && /:/{f=1} f{print} f&&/; Occupancy/{exit}' $f > /tmp/isa2/_k.s
grep "; NumVgprs\|; ScratchSize\|; TotalNumSgprs\|; Occupancy\|codeLen" /tmp/isa2/_k.s | tr '\n' ' '; echo
echo "VALU $(grep -c '^\s*v_' /tmp/isa2/_k.s) SALU $(grep -c '^\s*s_' /tmp/isa2/_k.s) readlane $(grep -c v_readlane /tmp/isa2/_k.s) writelane $(grep -c v_writelane /tmp/isa2/_k.s) s_load $(grep -c '^\s*s_load' /tmp/isa2/_k.s) scratch $(grep -c '^\s*scratch_' /tmp/isa2/_k.s)"
