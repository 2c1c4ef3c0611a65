#!/usr/bin/env python3
"""Re-wrap the prose of a markdown file to a column limit (tables, headings and fenced code stay as they are).
usage: wrap_md.py FILE [WIDTH=118]"""
import re, sys, textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out, block, fence = [], [], False
bullet = re.compile(r"^(\s*)([*-]|\d+\.)\s+")


def flush():
    global block
    if not block:
        return
    m = bullet.match(block[0])
    if m:
        first = m.group(0)
        rest = " " * len(first)
        text = " ".join([block[0][len(first):].strip()] + [l.strip() for l in block[1:]])
    else:
        first = rest = re.match(r"^\s*", block[0]).group(0)
        text = " ".join(l.strip() for l in block)
    out.extend(textwrap.wrap(text, width=width, initial_indent=first, subsequent_indent=rest, break_long_words=False, break_on_hyphens=False))
    block = []


for l in lines:
    if l.startswith("```"):
        flush(); fence = not fence; out.append(l); continue
    if fence or l.startswith("|") or l.startswith("#") or not l.strip():
        flush(); out.append(l); continue
    if bullet.match(l) and block:
        flush()
    block.append(l)
flush()
open(path, "w").write("\n".join(out))
