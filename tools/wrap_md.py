"""Re-fills the PROSE of a markdown file to a line width (default 118): paragraphs and list items whose lines exceed it.
Tables, headings, fenced / indented code and placeholder lines are left alone.  python tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 118
lines = open(path).read().split("\n")
out, i, fence = [], 0, False
item = re.compile(r"^(\s*)([*+-]|\d+\.|\(\d+\)|\([a-z]\))\s+")
while i < len(lines):
    l = lines[i]
    if l.startswith("```"):
        fence = not fence
    if fence or not l.strip() or l.startswith(("|", "#", "```", "<<", "    ")) and not item.match(l):
        out.append(l)
        i += 1
        continue
    # a block: this line + continuation lines (not blank, not a new item / table / heading / fence)
    m = item.match(l)
    first_prefix = m.group(0) if m else re.match(r"^\s*", l).group(0)
    block = [l]
    j = i + 1
    while j < len(lines) and lines[j].strip() and not lines[j].startswith(("|", "#", "```", "<<")) and not item.match(lines[j]):
        block.append(lines[j])
        j += 1
    if max(len(b) for b in block) > width + 4:
        hang = " " * len(first_prefix) if m else first_prefix
        if len(block) > 1 and m:
            cont = re.match(r"^\s*", block[1]).group(0)
            if len(cont) >= len(m.group(1)) + 1:
                hang = cont
        text = " ".join([block[0][len(first_prefix):].strip()] + [b.strip() for b in block[1:]])
        # keep double spaces after sentence ends out of the way of the filler
        filled = textwrap.fill(text, width=width, initial_indent=first_prefix, subsequent_indent=hang,
                               break_long_words=False, break_on_hyphens=False)
        out.extend(filled.split("\n"))
    else:
        out.extend(block)
    i = j
open(path, "w").write("\n".join(out))
