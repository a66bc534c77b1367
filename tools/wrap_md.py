#!/usr/bin/env python3
"""Wrap the prose of a Markdown file at 140 columns (tables, code blocks and headings are left alone; list items keep their hanging indent).
python tools/wrap_md.py DESIGN.md"""
import re
import sys
import textwrap

W = 140
path = sys.argv[1]
out, para, fence = [], [], False


def flush():
    if not para:
        return
    first = para[0]
    m = re.match(r"^(\s*(?:[*+-]|\d+\.)\s+)", first)
    indent = " " * len(m.group(1)) if m else re.match(r"^(\s*)", first).group(1)
    head = m.group(1) if m else indent
    text = " ".join(l.strip() for l in para)
    if m:
        text = text[len(m.group(1).strip()):].strip()
    out.extend(textwrap.wrap(text, W, initial_indent=head, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
    para.clear()


for line in open(path).read().split("\n"):
    if line.strip().startswith("```"):
        flush()
        fence = not fence
        out.append(line)
        continue
    if fence or line.startswith("|") or line.startswith("#") or not line.strip() or line.startswith("    "):
        flush()
        out.append(line)
        continue
    if re.match(r"^\s*(?:[*+-]|\d+\.)\s+", line) and para:
        flush()
    para.append(line)
flush()
open(path, "w").write("\n".join(out))
long_ = [i + 1 for i, l in enumerate(out) if len(l) > W and not l.startswith("|")]
print(f"{path}: {len(out)} lines, {len(long_)} non-table lines over {W} columns", long_[:10])
