#!/usr/bin/env python
"""Re-wrap the prose of a markdown file to <= 120 columns without changing its content: paragraphs and list items are re-flowed
(continuation lines of a list item are indented to its text), table rows, headings, fenced code and indented code are left alone.
usage: python scripts/wrap_md.py FILE..."""
import re
import sys
import textwrap

WIDTH = 120
ITEM = re.compile(r'^(\s*)([-*+]|\d+[.)])\s+')


def wrap_file(path):
    out, fence = [], False
    lines = open(path).read().split('\n')
    i = 0
    while i < len(lines):
        ln = lines[i]
        if ln.lstrip().startswith('```'):
            fence = not fence
            out.append(ln); i += 1; continue
        if fence or not ln.strip() or ln.lstrip().startswith(('|', '#', '>')) or ln.startswith('    ') and not ITEM.match(ln):
            out.append(ln); i += 1; continue
        # a paragraph or list item: this line plus its continuation lines (same block, not a new item / table / heading / blank)
        m = ITEM.match(ln)
        first_indent = m.group(0) if m else re.match(r'^\s*', ln).group(0)
        rest_indent = ' ' * len(first_indent) if m else first_indent
        text = ln[len(first_indent):]
        j = i + 1
        while j < len(lines):
            nx = lines[j]
            if not nx.strip() or nx.lstrip().startswith(('|', '#', '>', '```')) or ITEM.match(nx):
                break
            text += ' ' + nx.strip()
            j += 1
        if j == i + 1 and len(ln) <= WIDTH:
            out.append(ln)
        else:
            out.extend(textwrap.wrap(text, width=WIDTH, initial_indent=first_indent, subsequent_indent=rest_indent, break_long_words=False,
                                     break_on_hyphens=False) or [first_indent.rstrip()])
        i = j
    open(path, 'w').write('\n'.join(out))


for p in sys.argv[1:]:
    wrap_file(p)
