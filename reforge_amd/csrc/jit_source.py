#!/usr/bin/env python3
"""Build step: the device headers as ONE C++ raw string literal (build/rf_jit_source.inc), which rf_jit.cpp hands to
hiprtc.  Local #include lines are dropped (the files are concatenated in dependency order); the preprocessor
conditions that hide host-only parts under __HIPCC_RTC__ stay in the text."""
import re
import sys

out = []
for path in sys.argv[1:]:
    text = open(path).read()
    text = re.sub(r'(?m)^#include "[^"]+"[^\n]*\n', "", text)
    text = text.replace("#pragma once\n", "")
    out.append("// ---- %s\n%s" % (path, text))
body = "\n".join(out)
assert ')RFJIT"' not in body
sys.stdout.write('R"RFJIT(' + body + ')RFJIT"\n')
