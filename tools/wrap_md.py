#!/usr/bin/env python3
"""Re-wrap a Markdown file's prose to <= 120 columns (paragraphs and list items; headings, tables and code fences are kept
as they are) and, with --split, write every `### ` section of the file's `## 4.` chapter into design/<slug>.md, leaving an
index in its place.  usage: wrap_md.py [--split] FILE"""
import os
import re
import sys
import textwrap

W = 110
ITEM = re.compile(r"^(\s*)([*+-]|\d+\.)\s+")


def wrap_blocks(lines):
    out, i, n = [], 0, len(lines)
    while i < n:
        ln = lines[i]
        if ln.startswith("```"):
            j = i + 1
            while j < n and not lines[j].startswith("```"):
                j += 1
            out += lines[i:j + 1]
            i = j + 1
            continue
        if not ln.strip() or ln.startswith("#") or ln.lstrip().startswith("|"):
            out.append(ln)
            i += 1
            continue
        m = ITEM.match(ln)
        block = [ln.strip() if not m else ln[m.end():].strip()]
        j = i + 1
        while j < n:
            nx = lines[j]
            if (not nx.strip() or nx.startswith("#") or nx.startswith("```") or nx.lstrip().startswith("|")
                    or ITEM.match(nx)):
                break
            block.append(nx.strip())
            j += 1
        text = " ".join(block)
        if m:
            first = m.group(1) + m.group(2) + " "
            rest = " " * len(first)
        else:
            first = rest = re.match(r"^\s*", ln).group(0) if False else ""
        out += textwrap.wrap(text, W, initial_indent=first, subsequent_indent=rest, break_long_words=False,
                             break_on_hyphens=False)
        i = j
    return out


def slug(title):
    t = re.sub(r"`[^`]*`", lambda m: m.group(0).strip("`"), title)
    t = re.sub(r"[^A-Za-z0-9]+", "_", t).strip("_").lower()
    return t[:24].rstrip("_")


def main():
    split = "--split" in sys.argv
    path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    lines = open(path).read().split("\n")
    if split:
        start = next(i for i, l in enumerate(lines) if l.startswith("## 4."))
        end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("## "))
        chapter = lines[start:end]
        heads = [i for i, l in enumerate(chapter) if l.startswith("### ")]
        ddir = os.path.join(os.path.dirname(os.path.abspath(path)), "design")
        os.makedirs(ddir, exist_ok=True)
        index = chapter[:heads[0]] if heads else chapter
        index = [l for l in index]
        index += ["Each kernel family has a file of its own under `design/` (what it computes, the reference lines it replaces,",
                  "the roofline that bounds it, its algorithmic bytes / FLOPs per unit, what was measured and what was tried and",
                  "not kept):", ""]
        for k, h in enumerate(heads):
            stop = heads[k + 1] if k + 1 < len(heads) else len(chapter)
            title = chapter[h][4:].strip()
            name = f"{k + 1:02d}_{slug(title)}.md"
            body = ["# " + title, "", "(Part of DESIGN.md §4 — kernels.)", ""] + chapter[h + 1:stop]
            with open(os.path.join(ddir, name), "w") as fh:
                fh.write("\n".join(wrap_blocks(body)).rstrip("\n") + "\n")
            index.append(f"* [`design/{name}`](design/{name}) — {title}")
        index.append("")
        lines = lines[:start] + index + lines[end:]
    with open(path, "w") as fh:
        fh.write("\n".join(wrap_blocks(lines)).rstrip("\n") + "\n")


if __name__ == "__main__":
    main()
