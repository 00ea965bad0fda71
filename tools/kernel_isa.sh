#!/bin/bash
# usage: kernel_isa.sh file.s <mangled-name-substring> > kernel.s   -- cut one kernel's ISA out of a hipcc -S file
awk -v pat="$2" '$0 ~ "^_Z.*" pat ".*:" {on=1} on {print} on && /s_endpgm/ {exit}' "$1"
