#!/usr/bin/env python3
"""Reflow the paragraphs and list items of a Markdown file to a column limit (default 118), leaving headings, tables, block
quotes and fenced code as they are.  A list item keeps its marker and a hanging indent.   python tools/reflow_md.py FILE [COLS]"""
import re
import sys
import textwrap

path, cols = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path, encoding="utf-8").read().split("\n")
out, buf, fence = [], [], False


def flush():
    global buf
    if not buf:
        return
    first = buf[0]
    m = re.match(r"^(\s*)([*+-]|\d+\.)\s+", first)
    if m:
        indent0, hang = m.group(0), " " * len(m.group(0))
        text = first[len(indent0):] + " " + " ".join(s.strip() for s in buf[1:])
    else:
        lead = re.match(r"^\s*", first).group(0)
        indent0 = hang = lead
        text = " ".join(s.strip() for s in buf)
    text = re.sub(r"\s+", " ", text).strip()
    out.extend(textwrap.wrap(text, width=cols, initial_indent=indent0, subsequent_indent=hang, break_long_words=False,
                             break_on_hyphens=False))
    buf = []


for ln in lines:
    if ln.strip().startswith("```"):
        flush()
        fence = not fence
        out.append(ln)
        continue
    if fence or ln.startswith("#") or ln.startswith("|") or ln.startswith(">") or not ln.strip():
        flush()
        out.append(ln)
        continue
    if re.match(r"^\s*([*+-]|\d+\.)\s+", ln):   # a new list item starts a new block
        flush()
    buf.append(ln)
flush()
open(path, "w", encoding="utf-8").write("\n".join(out))
