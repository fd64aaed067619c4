"""(round 6) Re-wrap the prose of a Markdown file at 120 columns: paragraphs and list items only — fenced code, tables, headings, indented
code and lines that are one unbreakable token are left as they are; a list item keeps its hanging indent.    python tools/wrap_md.py FILE [width]"""
import re
import sys
import textwrap


def wrap(text, width=120):
    out, para, fence = [], [], False

    def flush():
        if not para:
            return
        first = para[0]
        m = re.match(r"^(\s*)((?:[-*+]|\d+\.)\s+)?", first)
        indent, bullet = m.group(1), m.group(2) or ""
        body = " ".join(l.strip() for l in para)
        if bullet:
            body = body[len(bullet):]
        hang = indent + " " * len(bullet)
        lines = textwrap.wrap(body, width=width, initial_indent=indent + bullet, subsequent_indent=hang, break_long_words=False,
                              break_on_hyphens=False)
        out.extend(lines or [first])
        para.clear()

    for line in text.split("\n"):
        s = line.strip()
        if s.startswith("```"):
            flush(); fence = not fence; out.append(line); continue
        if fence or s.startswith("|") or s.startswith("#") or line.startswith("    ") and not para or not s:
            flush(); out.append(line); continue
        # a new list item or a line with another indent than the running paragraph starts a new paragraph
        if para and (re.match(r"^\s*(?:[-*+]|\d+\.)\s+", line) or re.match(r"^(\s*)", line).group(1) != re.match(r"^(\s*)", para[-1]).group(1) and not re.match(r"^\s*(?:[-*+]|\d+\.)\s+", para[0])):
            flush()
        para.append(line)
    flush()
    return "\n".join(out)


if __name__ == "__main__":
    path, width = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 120
    src = open(path, encoding="utf-8").read()
    open(path, "w", encoding="utf-8").write(wrap(src, width))
