#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at 120 columns (tables, headings, code fences and the text inside them are left
alone; list items keep a hanging indent).  python tools/reflow_md.py DESIGN.md"""
import re
import sys
import textwrap

WIDTH = 120


def reflow(text):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", first)
        indent = m.group(1) or ""
        marker = m.group(2) or ""
        body = " ".join([first[len(indent) + len(marker):].strip()] + [l.strip() for l in para[1:]])
        hang = indent + " " * len(marker)
        out.extend(textwrap.wrap(body, WIDTH, initial_indent=indent + marker, subsequent_indent=hang, break_long_words=False,
                                 break_on_hyphens=False) or [indent + marker.rstrip()])
        para.clear()

    for line in text.split("\n"):
        s = line.rstrip()
        if s.lstrip().startswith("```"):
            flush()
            fence = not fence
            out.append(s)
            continue
        if fence or s.startswith("|") or s.startswith("#") or s.startswith("    ") and not para:
            flush()
            out.append(s)
            continue
        if not s.strip():
            flush()
            out.append("")
            continue
        if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", s) and para:  # a new list item ends the one before
            flush()
        para.append(s)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    for p in sys.argv[1:]:
        t = open(p).read()
        r = reflow(t)
        open(p, "w").write(r if r.endswith("\n") else r + "\n")
