#!/usr/bin/env python
"""Lists every `test "..."` of the reference files on the path and whether some test under tests/ cites its lines (file.zig:a-b).
Runs only where /root/reference exists (the build container); the output of the last run is kept in profiles/.
usage: python tools/reference_test_coverage.py [/root/reference]"""
import glob
import re
import sys

ROOT = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FILES = {  # citation key (file stem) -> path; two stems are shared by a source file and its test file: both are listed
    "filters": ["src/image/tests/filters.zig"], "interpolation": ["src/image/tests/interpolation.zig"], "resize": ["src/image/tests/resize.zig"],
    "transforms": ["src/image/tests/transforms.zig", "src/geometry/transforms.zig"], "integral": ["src/image/tests/integral.zig"],
    "binary": ["src/image/tests/binary.zig"], "shen_castan": ["src/image/tests/shen_castan.zig"], "border": ["src/image/border.zig"],
    "color": ["src/color.zig"], "pyramid": ["src/image/pyramid.zig"], "blending": ["src/blending.zig"], "png": ["src/codecs/png.zig"], "jpeg": ["src/codecs/jpeg.zig"],
}
# tests/test_oracle_sampling.py cites the lines of its section's file without repeating the stem ("# :36-70")
SECTION = re.compile(r"# ---- (?:[\w/]*/)?(\w+)\.zig")
OFF_PATH = {("color", 1585): "hex strings: not an image operation", ("color", 1775): "float colour blending outside Image.insert"}
cites = {}
for path in glob.glob("tests/*.py"):
    section = None
    for line in open(path).read().split("\n"):
        s = SECTION.search(line)
        if s:
            section = s.group(1)
        stem = section
        # "<stem>.zig:a-b" names the file; a bare ":a-b" (after a comma, a bracket or "# ") continues with the last file named
        for m in re.finditer(r"(?:(\w+)\.zig)?:(\d+)(?:-(\d+))?", line):
            if m.group(1):
                stem = m.group(1)
            elif m.start() == 0 or line[m.start() - 1] not in " (,":
                continue
            if stem:
                cites.setdefault(stem, []).append((int(m.group(2)), int(m.group(3) or m.group(2))))
total = missing = 0
for stem, paths in FILES.items():
    for path in paths:
        src = open(f"{ROOT}/{path}").read().split("\n")
        tests = [(i + 1, l[6:-3]) for i, l in enumerate(src) if l.startswith('test "')]
        ends = [t[0] for t in tests[1:]] + [len(src)]
        unc = [(ln, name) for (ln, name), end in zip(tests, ends) if not any(a <= end and b >= ln for a, b in cites.get(stem, []))]
        off = [(ln, name) for ln, name in unc if (stem, ln) in OFF_PATH]
        unc = [t for t in unc if t not in off]
        total += len(tests)
        missing += len(unc)
        total -= len(off)
        print(f"{path}: {len(tests)} tests, {len(tests) - len(unc) - len(off)} cited" + ("" if not off else "; off the path: " + "; ".join(f"{ln} {name} ({OFF_PATH[(stem, ln)]})" for ln, name in off))
              + ("" if not unc else "; NOT CITED: " + "; ".join(f"{ln} {name}" for ln, name in unc)))
print(f"TOTAL {total} reference tests on the path, {total - missing} cited by tests/, {missing} not")
