#!/bin/bash
# isa_kernel.sh <file.s> <mangled-name substring> : print one kernel's ISA (label .. .Lfunc_end) from a hipcc -S listing
awk -v pat="$2" '$0 ~ "^[_A-Za-z0-9]*"pat"[_A-Za-z0-9]*:" && !f {f=1} f{print} /^\.Lfunc_end/{if(f){exit}}' "$1"
