#!/usr/bin/env python3
"""Emit the device headers as C++ raw string literals (adjacent literals concatenate) for hiprtc."""
import os
import re
import sys

out = []
for path in sys.argv[1:]:
    text = open(path).read().replace("#pragma once", "")
    # hiprtc gets ONE string: textual includes of sibling files (#include "x.inc", device/qhip_agg_tile.inc) are pasted here
    here = os.path.dirname(path)
    text = re.sub(r'^#include "([A-Za-z0-9_]+\.inc)"[ \t]*$', lambda m: open(os.path.join(here, m.group(1))).read().rstrip("\n"), text, flags=re.M)
    # MSVC-free toolchain, but keep each literal comfortably small
    for k in range(0, len(text), 8000):
        out.append('R"QHIPSRC(' + text[k:k + 8000] + ')QHIPSRC"')
print("\n".join(out))
