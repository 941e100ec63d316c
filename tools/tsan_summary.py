#!/usr/bin/env python
"""Summarises a ThreadSanitizer log of tools/tsan_run.sh: for every report, the module of the first frame below the sanitizer's
own interceptor in each of the two racing accesses.  A report is about THIS library's code when one of those frames is in
libcilqr_hip_tsan.so or in the driver program; reports whose racing accesses both sit inside libamdhip64 / libhsa-runtime64
(not instrumented: their own locks and signals are invisible to the sanitizer) are listed by module only.
    python tools/tsan_summary.py <log>"""
import collections
import re
import sys

log = open(sys.argv[1]).read()
reports = log.split("WARNING: ThreadSanitizer:")[1:]
OURS = ("libcilqr_hip_tsan.so", "tsan_threads")


def top_module(block):
    for line in block.splitlines():
        m = re.match(r"\s+#\d+ .*\((\S+?)\+0x[0-9a-f]+\)", line)
        if not m:
            continue
        mod = m.group(1)
        if "libclang_rt" in mod:
            continue
        return mod, line.strip()
    return "?", ""


by_pair = collections.Counter()
ours = []
for r in reports:
    kind = r.split("(pid")[0].strip()
    blocks = [b for b in r.split("\n\n") if re.search(r"^\s+(Write|Read|Atomic|Previous)", b, re.M) or b.lstrip().startswith(("Write", "Read", "Atomic"))]
    tops = [top_module(b) for b in blocks[:2]]
    mods = tuple(sorted(t[0] for t in tops))
    by_pair[(kind, mods)] += 1
    if any(any(o in t[0] for o in OURS) for t in tops):
        ours.append((kind, tops))
ok = "tsan_threads ok" in log
print(f"driver finished and every result was bit-identical: {ok}")
print(f"ThreadSanitizer reports: {len(reports)}")
print(f"reports with a racing access in this library's code (libcilqr_hip_tsan.so / the driver): {len(ours)}")
for kind, tops in ours[:20]:
    print("  ", kind, "|", " || ".join(t[1] for t in tops))
print("by kind and by the modules of the two racing accesses (first frame below the interceptors):")
for (kind, mods), n in by_pair.most_common():
    print(f"  {n:5d}  {kind:24s} {' / '.join(mods)}")
