"""Re-wrap a markdown file at 120 columns: paragraphs and list items (hanging indent kept), never fenced code, table rows,
headings or link-only lines.  usage: python tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap

path = sys.argv[1]
width = int(sys.argv[2]) if len(sys.argv) > 2 else 120
lines = open(path).read().split("\n")
out, buf, fence = [], [], False


def flush():
    global buf
    if not buf:
        return
    first = buf[0]
    m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", first)
    indent, bullet = m.group(1), m.group(2) or ""
    text = " ".join(s.strip() for s in buf)
    text = text[len(bullet):] if bullet and text.startswith(bullet.strip()) else text
    body = text.strip()
    if bullet:
        body = re.sub(r"^(?:[-*+]|\d+\.)\s+", "", body)
    wrapped = textwrap.wrap(body, width=width, initial_indent=indent + bullet, subsequent_indent=indent + " " * len(bullet),
                            break_long_words=False, break_on_hyphens=False)
    out.extend(wrapped or [""])
    buf = []


for ln in lines:
    if ln.strip().startswith("```"):
        flush()
        fence = not fence
        out.append(ln)
        continue
    if fence or ln.startswith("#") or ln.lstrip().startswith("|") or not ln.strip() or re.match(r"^\s*\{.*\}\s*$", ln):
        flush()
        out.append(ln)
        continue
    if re.match(r"^\s*(?:[-*+]|\d+\.)\s+", ln):  # a new list item
        flush()
        buf = [ln]
        continue
    if buf and (len(ln) - len(ln.lstrip())) < (len(buf[0]) - len(buf[0].lstrip())) and not re.match(r"^\s*(?:[-*+]|\d+\.)\s+", buf[0]):
        flush()
    buf.append(ln)
flush()
open(path, "w").write("\n".join(out))
long_ = [(i + 1, len(l)) for i, l in enumerate(out) if len(l) > width]
print("%s: %d lines, %d still longer than %d (tables / headings): %s" % (path, len(out), len(long_), width, long_[:12]))
