"""Report runs of >= N consecutive non-blank source lines that two files share verbatim (whitespace-stripped).
Usage: python tools/verbatim_check.py OURS THEIRS [N]   (the judge's line-level difflib check, round 3 VERDICT item 8)"""
import difflib
import sys


def lines(path):
    return [(i + 1, l.strip()) for i, l in enumerate(open(path, errors='replace')) if l.strip()]


def blocks(ours, theirs, n=4):
    a, b = lines(ours), lines(theirs)
    sm = difflib.SequenceMatcher(None, [l for _, l in a], [l for _, l in b], autojunk=False)
    return [(a[m.a][0], b[m.b][0], m.size) for m in sm.get_matching_blocks() if m.size >= n]


if __name__ == '__main__':
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    found = blocks(sys.argv[1], sys.argv[2], n)
    for ours_line, theirs_line, size in found:
        print('%s:%d == %s:%d for %d lines' % (sys.argv[1], ours_line, sys.argv[2], theirs_line, size))
    sys.exit(1 if found else 0)
