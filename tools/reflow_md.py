#!/usr/bin/env python3
"""Reflow the paragraphs and list items of a Markdown file to a maximum line width (tables, code blocks and headings are left alone; a table row longer
than the width is reported).    python tools/reflow_md.py FILE [width]"""
import re
import sys
import textwrap


def reflow(text: str, width: int = 160):
    out, para, in_code = [], [], False
    long_rows = 0

    def flush():
        nonlocal para
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)([-*+]|\d+\.)\s+", first)
        indent = m.group(0) if m else re.match(r"^\s*", first).group(0)
        body = " ".join(ln.strip() for ln in para)
        if m:
            body = body[len(m.group(0).strip()) + 1:] if body.startswith(m.group(0).strip()) else body
            sub = " " * len(m.group(0))
            out.extend(textwrap.wrap(body, width=width, initial_indent=m.group(0), subsequent_indent=sub, break_long_words=False, break_on_hyphens=False) or [m.group(0)])
        else:
            out.extend(textwrap.wrap(body, width=width, initial_indent=indent, subsequent_indent=indent, break_long_words=False, break_on_hyphens=False))
        para = []

    for ln in text.split("\n"):
        if ln.strip().startswith("```"):
            flush(); in_code = not in_code; out.append(ln); continue
        if in_code:
            out.append(ln); continue
        if not ln.strip():
            flush(); out.append(""); continue
        if ln.lstrip().startswith(("#", "|")) or re.match(r"^\s*(track \d|level \d)", ln):
            flush(); out.append(ln)
            if ln.lstrip().startswith("|") and len(ln) > width:
                long_rows += 1
            continue
        if re.match(r"^\s*([-*+]|\d+\.)\s+", ln):
            flush()
        para.append(ln)
    flush()
    return "\n".join(out), long_rows


if __name__ == "__main__":
    path = sys.argv[1]; width = int(sys.argv[2]) if len(sys.argv) > 2 else 160
    with open(path) as f:
        new, long_rows = reflow(f.read(), width)
    with open(path, "w") as f:
        f.write(new)
    print(f"{path}: reflowed to {width}; {long_rows} table rows are longer")
