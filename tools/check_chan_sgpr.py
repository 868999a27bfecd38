#!/usr/bin/env python3
"""Build-time check for csrc/scan_fwd_chan.hip: the hand-scheduled token blocks keep B/C rows in FIXED SGPRs
(X = s[68:99], Y = s[36:67]) across compiler-generated code, which is only sound if the compiler's own code
  (1) never touches s68 or above, and
  (2) never touches s36..s67 while Y is live (from the `lo_x` block that loads it to the end of the `hi_y` block).
Usage: python tools/check_chan_sgpr.py <file.s>   (device assembly from hipcc -S --cuda-device-only)"""
import re, sys

def sgprs(line):
    out = set()
    for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", line):
        out.update(range(int(a), int(b) + 1))
    out.update(int(x) for x in re.findall(r"\bs(\d+)\b", line))
    return out

bad, kernels, pairs = [], 0, {}
name, in_asm, y_live, tag = None, False, False, None
for ln, line in enumerate(open(sys.argv[1]), 1):
    m = re.match(r"^(_ZN5vivim19ssm_fwd_chan_kernel\S*):", line)
    if m:
        name, in_asm, y_live, kernels = m.group(1), False, False, kernels + 1
        pairs[name] = {}
        continue
    if name is None:
        continue
    if re.match(r"^\.Lfunc_end|^\s+\.end_amdhsa_kernel|^[A-Za-z_][\w.$]*:\s*$", line):   # end of the function's text (an early
        if y_live:                                                                        # s_endpgm does not end it)
            bad.append((name, ln, "Y still live at the end of the kernel", line.strip()))
        name = None
        continue
    if "ASMSTART" in line:
        in_asm, tag = True, None
        continue
    if "ASMEND" in line:
        in_asm = False
        if tag == "lo_x":
            y_live = True
        elif tag == "hi_y":
            y_live = False
        continue
    if in_asm:
        t = re.search(r"; CHAN (\w+)", line)
        if t:
            tag = t.group(1)
            pairs[name][tag] = pairs[name].get(tag, 0) + 1
        continue
    code = line.split(";")[0]
    if not re.match(r"^\s+[sv]_|^\s+(ds|global|buffer|flat|scratch)_", code):
        continue
    regs = sgprs(code)
    if any(r >= 68 and r <= 101 for r in regs):
        bad.append((name, ln, "touches X (s68+)", line.strip()))
    if y_live and any(36 <= r <= 67 for r in regs):
        bad.append((name, ln, "touches Y (s36..s67) while it is live", line.strip()))
assert kernels > 0, "no ssm_fwd_chan_kernel found in " + sys.argv[1]
for k, tags in pairs.items():          # every block that makes Y live (lo_x) has the block that ends its life (hi_y), and vice versa
    if tags.get("lo_x", 0) == 0 or tags.get("lo_x", 0) != tags.get("hi_y", 0) or tags.get("lo_y", 0) != tags.get("hi_x", 0):
        bad.append((k, 0, "unbalanced token blocks", str(tags)))
for b in bad[:20]:
    print("check-chan-sgpr: %s line %d: %s: %s" % b)
if bad:
    sys.exit(1)
print("check-chan-sgpr: %d kernels, the compiler stays clear of the reserved scalar sets" % kernels)
