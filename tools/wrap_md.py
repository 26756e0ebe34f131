"""Re-wraps the paragraphs and list items of markdown files at 118 columns (tables, headings, code blocks and blank lines stay
as they are): python tools/wrap_md.py FILE..."""
import re
import sys
import textwrap

W = 118


def flush(block, out):
    if not block:
        return
    first = block[0]
    m = re.match(r"^(\s*)((?:[*\-]|\d+\.)\s+)?", first)
    indent, bullet = m.group(1), m.group(2) or ""
    text = " ".join(l.strip() for l in block)
    if bullet:
        text = text[len(bullet.strip()):].strip()
    sub = indent + " " * len(bullet)
    out.extend(textwrap.wrap(text, width=W, initial_indent=indent + bullet, subsequent_indent=sub, break_long_words=False,
                             break_on_hyphens=False))
    block.clear()


for path in sys.argv[1:]:
    out, block, code = [], [], False
    for line in open(path).read().split("\n"):
        if line.startswith("```"):
            flush(block, out)
            code = not code
            out.append(line)
        elif code or line.startswith(("|", "#", ">")) or not line.strip():
            flush(block, out)
            out.append(line)
        elif re.match(r"^\s*([*\-]|\d+\.)\s+", line):
            flush(block, out)
            block.append(line)
        else:
            block.append(line)
    flush(block, out)
    open(path, "w").write("\n".join(out))
