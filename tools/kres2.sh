#!/bin/bash
# one line per kernel of a .res file: name, VGPRs, SGPR spills, VGPR spills, scratch bytes, occupancy
awk '/Function Name:/{name=$NF} /  VGPRs:/{v=$(NF-1)} /SGPRs Spill:/{ss=$(NF-1)} /VGPRs Spill:/{vs=$(NF-1)} /ScratchSize/{sc=$(NF-1)} /Occupancy/{oc=$(NF-1)} /LDS Size/{printf "%-90s vgpr %3s  sgpr-spill %3s  vgpr-spill %3s  scratch %4s  occ %s\n", name, v, ss, vs, sc, oc}' "$@"
