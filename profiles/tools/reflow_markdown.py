import re, sys, textwrap
W = 148
src = open(sys.argv[1]).read().split('\n')
out = []
def split_row(r):
    cells = [c.strip() for c in re.split(r'(?<!\\)\|', r.strip())]
    if cells and cells[0] == '': cells = cells[1:]
    if cells and cells[-1] == '': cells = cells[:-1]
    return cells
def wrap_block(first_indent, rest_indent, text):
    return textwrap.wrap(text, width=W, initial_indent=first_indent, subsequent_indent=rest_indent, break_long_words=False, break_on_hyphens=False)
BUL = re.compile(r'^(\s*)([*\-+] |\d+\. )(.*)$')
def special(l):
    return l.strip() == '' or l.startswith('#') or l.strip().startswith('```') or l.lstrip().startswith('|')
i = 0
in_code = False
while i < len(src):
    line = src[i]
    if line.strip().startswith('```'):
        in_code = not in_code; out.append(line); i += 1; continue
    if in_code or line.strip() == '' or line.startswith('#'):
        out.append(line); i += 1; continue
    if line.lstrip().startswith('|'):
        j = i; rows = []
        while j < len(src) and src[j].lstrip().startswith('|'):
            rows.append(src[j]); j += 1
        is_table = len(rows) >= 2 and re.match(r'^\s*\|[\s:\-|]+\|\s*$', rows[1])
        if is_table and max(len(r.encode()) for r in rows) <= 160:
            out += rows
        elif is_table:
            hdr = split_row(rows[0])
            for r in rows[2:]:
                cells = split_row(r)
                parts = [((f'*{h}*: ' if h else '') + c) for h, c in zip(hdr[1:], cells[1:]) if c]
                out += wrap_block('* ', '  ', f'**{cells[0]}**' + (' — ' + ' · '.join(parts) if parts else ''))
        else:
            for r in rows:
                cells = split_row(r)
                out += wrap_block('* ', '  ', f'**{cells[0]}**' + (' — ' + ' · '.join(c for c in cells[1:] if c) if len(cells) > 1 else ''))
        i = j; continue
    # paragraph or list item: gather continuation lines
    m = BUL.match(line)
    if m:
        ind, mark, text = m.group(1), m.group(2), m.group(3)
        first, rest = ind + mark, ind + ' ' * len(mark)
    else:
        ind = re.match(r'^(\s*)', line).group(1)
        first = rest = ind; text = line.strip()
    j = i + 1
    while j < len(src) and not special(src[j]) and not BUL.match(src[j]):
        text += ' ' + src[j].strip(); j += 1
    block = src[i:j]
    if max(len(b.encode()) for b in block) <= 160:
        out += block            # already fine: keep the author's line breaks
    else:
        out += wrap_block(first, rest, text)
    i = j
open(sys.argv[1], 'w').write('\n'.join(out))
