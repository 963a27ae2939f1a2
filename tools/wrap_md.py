#!/usr/bin/env python3
"""Re-wrap the prose of a Markdown file at 120 columns (tables, code fences and headings are left alone; list items keep a hanging
indent).   python tools/wrap_md.py DESIGN.md"""
import re
import sys
import textwrap

WIDTH = 120


def wrap_file(path):
    lines = open(path).read().split("\n")
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[*\-+]|\d+\.)\s+)?", first)
        lead, bullet = m.group(1), m.group(2) or ""
        text = " ".join(l.strip() for l in para)
        text = text[len(bullet):] if bullet and text.startswith(bullet.strip()) else text
        text = text.lstrip()
        if bullet:
            text = text[len(bullet.strip()):].lstrip() if text.startswith(bullet.strip()) else text
        out.extend(textwrap.wrap(text, WIDTH, initial_indent=lead + bullet, subsequent_indent=lead + " " * len(bullet),
                                 break_long_words=False, break_on_hyphens=False) or [""])
        para.clear()

    for l in lines:
        if l.strip().startswith("```"):
            flush(); fence = not fence; out.append(l); continue
        if fence or l.startswith("|") or l.startswith("#") or not l.strip():
            flush(); out.append(l); continue
        # a new list item starts a new paragraph
        if re.match(r"^\s*(?:[*\-+]|\d+\.)\s+", l) and para:
            flush()
        para.append(l)
    flush()
    open(path, "w").write("\n".join(out))


if __name__ == "__main__":
    for p in sys.argv[1:]:
        wrap_file(p)
