"""Re-flow the prose of a markdown file at ~160 columns: paragraphs and list items (with their continuation lines) are joined and wrapped; tables, headings, code
fences, html and blank lines are left alone."""
import re, sys, textwrap
W = 160
BUL = re.compile(r"(\s*)((?:[*\-+]|\d+\.)\s+)")
def special(l): return not l.strip() or l.lstrip().startswith(("|", "#", "```", "<", ">")) or l.startswith("    ") and not BUL.match(l)
for path in sys.argv[1:]:
    lines, out, i, fence = open(path).read().split("\n"), [], 0, False
    while i < len(lines):
        l = lines[i]
        if l.strip().startswith("```"): fence = not fence
        if fence or special(l): out.append(l); i += 1; continue
        m = BUL.match(l)
        lead, bullet = (m.group(1), m.group(2)) if m else (re.match(r"\s*", l).group(0), "")
        body = [l[len(lead) + len(bullet):].strip()]
        i += 1
        while i < len(lines) and not special(lines[i]) and not BUL.match(lines[i]) and not lines[i].strip().startswith("```"):
            # a continuation line of a list item is indented; a paragraph's is not: anything else starts a new block
            if bullet and not lines[i].startswith(" "): break
            body.append(lines[i].strip()); i += 1
        text = " ".join(body)
        wrapped = textwrap.wrap(text, width=W - len(lead) - len(bullet), break_long_words=False, break_on_hyphens=False) or [""]
        out.append(lead + bullet + wrapped[0])
        out += [lead + " " * len(bullet) + w for w in wrapped[1:]]
    open(path, "w").write("\n".join(out))
