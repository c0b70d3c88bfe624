/*
 * sre_parser.c — hand-written lexer + recursive-descent parser for the regex
 * dialect of the reference (whose front end is a bison grammar + hand lexer,
 * reference src/sregex/sre_yyparser.y:105-345 and :350-1795).  bison is not
 * part of this build; this file reproduces the *language* — token rules, AST
 * shapes and syntax-error offsets — and is pinned against the reference by the
 * AST dumps / error offsets of all 1999 reference test blocks in tests/golden.
 *
 * Grammar (LL(1) restatement of sre_yyparser.y:105-345):
 *   regex  := alt EOF
 *   alt    := concat ('|' concat)*                    left-nested ALT
 *   concat := repeat*                                 empty => NIL, else left-nested CAT
 *   repeat := atom [ ('*' | '+' | '?' | CQUANT) ['?'] ]
 *   atom   := '(' alt ')' | '(' '?' ':' alt ')' | CHAR | '.' | '^' | '$'
 *           | ASSERTION | CHAR_CLASS | ':'
 * A syntax error is reported at the first byte of the first token that cannot
 * continue a valid parse (sre_yyparser.y:1798-1803).
 */
#include "sre_program.h"
#include <stdio.h>
#include <string.h>

/* ------------------------------------------------------------------ tokens */

enum {
    T_EOF = 256, T_BAD, T_CHAR, T_CQUANT, T_CLASS, T_ASSERT
    /* operators are returned as their own byte value: | * + ? ( ) : . ^ $ */
};

typedef struct {
    sre_pool_t     *pool;
    const uint8_t  *src;      /* start of the NUL-terminated regex */
    const uint8_t  *p;        /* read cursor */
    int             flags;
    /* current token */
    int             tok;
    const uint8_t  *tok_pos;
    uint8_t         ch;       /* T_CHAR */
    int             qfrom, qto;   /* T_CQUANT, qto == -1 for {n,} */
    sre_regex_t    *node;     /* T_CLASS / T_ASSERT */
    /* parser state */
    sre_uint_t     *ncaps;
    int             oom;
} sre_lex_t;

/* byte ranges of the class escapes (sre_yyparser.y:358-384); pairs from,to */
static const uint8_t rg_d[] = { '0', '9' };
static const uint8_t rg_D[] = { 0, 47, 58, 255 };
static const uint8_t rg_w[] = { 'A', 'Z', 'a', 'z', '0', '9', '_', '_' };
static const uint8_t rg_W[] = { 0, 47, 58, 64, 91, 94, 96, 96, 123, 255 };
static const uint8_t rg_s[] = { ' ', ' ', '\f', '\f', '\n', '\n', '\r', '\r', '\t', '\t' };
static const uint8_t rg_S[] = { 0, 8, 11, 11, 14, 31, 33, 255 };
static const uint8_t rg_h[] = { 0x09, 0x09, 0x20, 0x20, 0xa0, 0xa0 };
static const uint8_t rg_H[] = { 0x00, 0x08, 0x0a, 0x1f, 0x21, 0x9f, 0xa1, 0xff };
static const uint8_t rg_v[] = { 0x0a, 0x0a, 0x0b, 0x0b, 0x0c, 0x0c, 0x0d, 0x0d, 0x85, 0x85 };
static const uint8_t rg_V[] = { 0x00, 0x09, 0x0e, 0x84, 0x86, 0xff };
static const uint8_t rg_nl[] = { '\n', '\n' };

#define RG(a)  a, (unsigned) sizeof(a)

static int is_print(unsigned c) { return c >= 0x20 && c <= 0x7e; }  /* "C" locale */
static int is_oct(unsigned c) { return c >= '0' && c <= '7'; }
static int is_dig(unsigned c) { return c >= '0' && c <= '9'; }

static int
hex_val(unsigned c)
{
    if (c >= '0' && c <= '9') return (int) c - '0';
    if (c >= 'A' && c <= 'F') return (int) c - 'A' + 10;
    if (c >= 'a' && c <= 'f') return (int) c - 'a' + 10;
    return -1;
}

/* ------------------------------------------------------------------ AST */

static sre_regex_t *
node_new(sre_lex_t *lx, sre_re_type_t type, sre_regex_t *l, sre_regex_t *r)
{
    sre_regex_t *n = sre_pcalloc(lx->pool, sizeof(sre_regex_t));
    if (n == NULL) {
        lx->oom = 1;
        return NULL;
    }
    n->type = type;
    n->left = l;
    n->right = r;
    return n;
}

static int
ranges_insert(sre_lex_t *lx, sre_rangevec_t *v, uint32_t at, unsigned from, unsigned to)
{
    if (v->n == v->cap) {
        uint32_t ncap = v->cap ? v->cap * 2 : 8;
        sre_range_t *nr = sre_palloc(lx->pool, ncap * sizeof(sre_range_t));
        if (nr == NULL) {
            lx->oom = 1;
            return -1;
        }
        if (v->n) {
            memcpy(nr, v->r, v->n * sizeof(sre_range_t));
        }
        v->r = nr;
        v->cap = ncap;
    }
    memmove(&v->r[at + 1], &v->r[at], (v->n - at) * sizeof(sre_range_t));
    v->r[at].from = (uint8_t) from;
    v->r[at].to = (uint8_t) to;
    v->n++;
    return 0;
}

static int
ranges_push(sre_lex_t *lx, sre_rangevec_t *v, unsigned from, unsigned to)
{
    return ranges_insert(lx, v, v->n, from, to);
}

static int
ranges_push_table(sre_lex_t *lx, sre_rangevec_t *v, const uint8_t *tab, unsigned nbytes)
{
    for (unsigned i = 0; i + 1 < nbytes; i += 2) {
        if (ranges_push(lx, v, tab[i], tab[i + 1]) != 0) return -1;
    }
    return 0;
}

/* ASCII case folding of a class: after every range that overlaps A-Z (resp.
 * a-z) insert the overlap shifted into the other case (reference
 * sre_regex.c:170-214; the inserted ranges are not re-examined). */
static int
ranges_make_caseless(sre_lex_t *lx, sre_rangevec_t *v)
{
    for (uint32_t i = 0; i < v->n; i++) {
        unsigned from = v->r[i].from, to = v->r[i].to;
        if (to >= 'A' && from <= 'Z') {
            unsigned f = (from > 'A' ? from : 'A') + 32, t = (to < 'Z' ? to : 'Z') + 32;
            if (ranges_insert(lx, v, ++i, f, t) != 0) return -1;
        }
        if (to >= 'a' && from <= 'z') {
            unsigned f = (from > 'a' ? from : 'a') - 32, t = (to < 'z' ? to : 'z') - 32;
            if (ranges_insert(lx, v, ++i, f, t) != 0) return -1;
        }
    }
    return 0;
}

static sre_regex_t *
class_node(sre_lex_t *lx, sre_re_type_t type, const uint8_t *tab, unsigned nbytes)
{
    sre_regex_t *n = node_new(lx, type, NULL, NULL);
    if (n && ranges_push_table(lx, &n->ranges, tab, nbytes) != 0) {
        return NULL;
    }
    return n;
}

static sre_regex_t *
any_node(sre_lex_t *lx)
{
    /* '.' and \C: every byte, or [^\n] under SRE_REGEX_NEWLINE
     * (sre_yyparser.y:293-297, 865-869) */
    if (lx->flags & SRE_REGEX_NEWLINE) {
        return class_node(lx, SRE_RE_NCLASS, RG(rg_nl));
    }
    return node_new(lx, SRE_RE_DOT, NULL, NULL);
}

/* ------------------------------------------------------------------ lexer */

static int tok_char(sre_lex_t *lx, unsigned c) { lx->ch = (uint8_t) c; return T_CHAR; }

static int
tok_assert(sre_lex_t *lx, unsigned bit)
{
    sre_regex_t *n = node_new(lx, SRE_RE_ASSERT, NULL, NULL);
    if (n == NULL) return T_BAD;
    n->assertion = (uint8_t) bit;
    lx->node = n;
    return T_ASSERT;
}

static int
tok_class(sre_lex_t *lx, sre_re_type_t type, const uint8_t *tab, unsigned nbytes)
{
    lx->node = class_node(lx, type, tab, nbytes);
    return lx->node ? T_CLASS : T_BAD;
}

/* \o{...}: up to 3 octal digits.  `strict` (inside [...]) rejects any other
 * byte before '}' (sre_yyparser.y:1168-1208); outside a class a foreign byte
 * ends the escape and is re-read as the next token (:474-531). */
static int
lex_oct_braced(sre_lex_t *lx, int strict, unsigned *out)
{
    unsigned c = *lx->p++, num = 0, i = 0;
    if (c != '{') return -1;
    c = *lx->p++;
    for (;;) {
        if (is_oct(c)) {
            num = (num << 3) + (c - '0');
        } else if (c == '}') {
            *out = num & 0xff;
            return 0;
        } else if (c == '\0' || strict) {
            return -1;
        } else {
            lx->p--;
            break;
        }
        if (++i == 3) {
            if (*lx->p++ != '}') return -1;
            if (num > 255) return -1;
            break;
        }
        c = *lx->p++;
    }
    *out = num & 0xff;
    return 0;
}

/* \xH, \xHH, \x{H}, \x{HH} (sre_yyparser.y:533-593, in class :1210-1262) */
static int
lex_hex(sre_lex_t *lx, int in_class, unsigned *out)
{
    unsigned c = *lx->p++, num = 0, i = 0;
    int braced = 0;
    if (c == '{') {
        braced = 1;
        c = *lx->p++;
    }
    for (;;) {
        int h = hex_val(c);
        if (h >= 0) {
            num = (num << 4) + (unsigned) h;
        } else if (braced) {
            if (c != '}') return -1;
            *out = num & 0xff;
            return 0;
        } else if (in_class && c == '\0') {
            return -1;
        } else {
            lx->p--;
            break;
        }
        if (++i == 2) {
            if (braced && *lx->p++ != '}') return -1;
            break;
        }
        c = *lx->p++;
    }
    *out = num & 0xff;
    return 0;
}

static int
lex_escape(sre_lex_t *lx)
{
    unsigned c = *lx->p++, num, i;

    if (c == '\0') return T_BAD;
    if (!is_print(c)) return tok_char(lx, c);
    if (strchr("'\" iM%@!,_-|*+?():.^$&\\/[]{}", (int) c)) return tok_char(lx, c);

    if (is_oct(c)) {
        /* \0, \dd, \ddd; a lone non-zero digit is a back-reference in Perl
         * and a syntax error here (sre_yyparser.y:419-453) */
        num = c - '0';
        i = 1;
        for (;;) {
            c = *lx->p;
            if (!is_oct(c)) {
                if (++i != 3 && num != 0) return T_BAD;
                return tok_char(lx, num);
            }
            num = (num << 3) + (c - '0');
            lx->p++;
            if (++i == 3) {
                if (num > 255) return T_BAD;
                return tok_char(lx, num);
            }
        }
    }

    switch (c) {
    case 'c':
        c = *lx->p++;
        if (c == '\0') return T_BAD;
        if (c >= 'a' && c <= 'z') c -= 32;
        return tok_char(lx, (c ^ 64) & 0xff);
    case 'o':
        if (lex_oct_braced(lx, 0, &num) != 0) return T_BAD;
        return tok_char(lx, num);
    case 'x':
        if (lex_hex(lx, 0, &num) != 0) return T_BAD;
        return tok_char(lx, num);

    case 'B': return tok_assert(lx, SRE_ASSERT_BIG_B);
    case 'b': return tok_assert(lx, SRE_ASSERT_SMALL_B);
    case 'z': return tok_assert(lx, SRE_ASSERT_SMALL_Z);
    case 'A': return tok_assert(lx, SRE_ASSERT_BIG_A);

    /* negated escapes are NCLASS over the positive table (:648-1021) */
    case 'd': return tok_class(lx, SRE_RE_CLASS, RG(rg_d));
    case 'D': return tok_class(lx, SRE_RE_NCLASS, RG(rg_d));
    case 'w': return tok_class(lx, SRE_RE_CLASS, RG(rg_w));
    case 'W': return tok_class(lx, SRE_RE_NCLASS, RG(rg_w));
    case 's': return tok_class(lx, SRE_RE_CLASS, RG(rg_s));
    case 'S': return tok_class(lx, SRE_RE_NCLASS, RG(rg_s));
    case 'h': return tok_class(lx, SRE_RE_CLASS, RG(rg_h));
    case 'H': return tok_class(lx, SRE_RE_NCLASS, RG(rg_h));
    case 'v': return tok_class(lx, SRE_RE_CLASS, RG(rg_v));
    case 'V': return tok_class(lx, SRE_RE_NCLASS, RG(rg_v));
    case 'N': return tok_class(lx, SRE_RE_NCLASS, RG(rg_nl));
    case 'C':
        lx->node = any_node(lx);
        return lx->node ? T_CLASS : T_BAD;

    case 't': return tok_char(lx, '\t');
    case 'n': return tok_char(lx, '\n');
    case 'r': return tok_char(lx, '\r');
    case 'f': return tok_char(lx, '\f');
    case 'a': return tok_char(lx, 7);
    case 'e': return tok_char(lx, 27);
    case '#': return tok_char(lx, c);
    default:
        return T_BAD;
    }
}

/* One escape inside [...].  Returns 0 and *out = byte for a single-byte
 * escape, 1 after appending a class escape's ranges, -1 on error. */
static int
lex_class_escape(sre_lex_t *lx, sre_rangevec_t *v, unsigned *pseen_dash, unsigned *out)
{
    unsigned c = *lx->p++, num, i;

    if (is_oct(c)) {
        /* no back-references inside a class: 1-3 digits (:1133-1156) */
        num = c - '0';
        i = 1;
        for (;;) {
            c = *lx->p;
            if (!is_oct(c)) break;
            num = (num << 3) + (c - '0');
            lx->p++;
            if (++i == 3) {
                if (num > 255) return -1;
                break;
            }
        }
        *out = num & 0xff;
        return 0;
    }

    switch (c) {
    case 'c':
        c = *lx->p++;
        if (c == '\0') return -1;
        if (c >= 'a' && c <= 'z') c -= 32;
        *out = (c ^ 64) & 0xff;
        return 0;
    case 'o':
        return lex_oct_braced(lx, 1, out);
    case 'x':
        return lex_hex(lx, 1, out);
    case 't': *out = '\t'; return 0;
    case 'n': *out = '\n'; return 0;
    case 'r': *out = '\r'; return 0;
    case 'f': *out = '\f'; return 0;
    case 'a': *out = 7; return 0;
    case 'e': *out = 27; return 0;
    case 'b': *out = 8; return 0;
    case '\0': return -1;
    case '#': *out = c; return 0;
    default: break;
    }

    if (!is_print(c) || strchr("'\" iMzC%@!,_-|*+?():.^$&\\/[]{}", (int) c)) {
        *out = c;
        return 0;
    }

    /* a class escape: a pending "x-" turns its dash into a literal (:1296-1314) */
    if (*pseen_dash) {
        if (ranges_push(lx, v, '-', '-') != 0) return -1;
        *pseen_dash = 0;
    }

    {
        const uint8_t *tab;
        unsigned nb;
        switch (c) {
        case 'd': tab = rg_d; nb = sizeof(rg_d); break;
        case 'D': tab = rg_D; nb = sizeof(rg_D); break;
        case 'w': tab = rg_w; nb = sizeof(rg_w); break;
        case 'W': tab = rg_W; nb = sizeof(rg_W); break;
        case 's': tab = rg_s; nb = sizeof(rg_s); break;
        case 'S': tab = rg_S; nb = sizeof(rg_S); break;
        case 'v': tab = rg_v; nb = sizeof(rg_v); break;
        case 'V': tab = rg_V; nb = sizeof(rg_V); break;
        case 'h': tab = rg_h; nb = sizeof(rg_h); break;
        case 'H': tab = rg_H; nb = sizeof(rg_H); break;
        default: return -1;
        }
        if (ranges_push_table(lx, v, tab, nb) != 0) return -1;
    }
    return 1;
}

/* [...] (sre_yyparser.y:1069-1670): leading '^' negates; ']' first is a
 * literal; '-' makes a range only between two single bytes; a trailing or
 * otherwise unusable '-' is a literal; reversed range / unterminated => error. */
static int
lex_bracket(sre_lex_t *lx)
{
    sre_regex_t    *n;
    sre_rangevec_t *v;
    unsigned        seen_dash = 0, no_dash = 0, count = 0, c;

    if (*lx->p == '^') {
        lx->p++;
        n = node_new(lx, SRE_RE_NCLASS, NULL, NULL);
    } else {
        n = node_new(lx, SRE_RE_CLASS, NULL, NULL);
    }
    if (n == NULL) return T_BAD;
    v = &n->ranges;

    for (;;) {
        int literal = 0;
        count++;
        c = *lx->p++;

        if (c == '\0') {
            return T_BAD;

        } else if (c == ']' && count > 1) {
            if (seen_dash && ranges_push(lx, v, '-', '-') != 0) return T_BAD;
            lx->node = n;
            return T_CLASS;

        } else if (c == '\\') {
            int rc = lex_class_escape(lx, v, &seen_dash, &c);
            if (rc < 0) return T_BAD;
            if (rc == 1) {
                no_dash = 1;
                continue;
            }
            literal = 1;

        } else if (c == '-') {
            if (!seen_dash && v->n && !no_dash) {
                seen_dash = 1;
                continue;
            }
            literal = 1;

        } else {
            literal = 1;
        }

        if (literal) {
            if (seen_dash) {
                sre_range_t *last = &v->r[v->n - 1];
                last->to = (uint8_t) c;
                if (last->to < last->from) return T_BAD;
                seen_dash = 0;
                no_dash = 1;
                continue;
            }
            no_dash = 0;
            if (ranges_push(lx, v, c, c) != 0) return T_BAD;
        }
    }
}

/* {n} {n,} {n,m} with n,m < 500 and n <= m; anything else leaves '{' a
 * literal (sre_yyparser.y:1693-1784).  {0,1} {0,} {1,} lex as ? * + */
static int
lex_brace(sre_lex_t *lx)
{
    const uint8_t *q = lx->p;
    long           from = 0, to;
    unsigned       c = *q;

    if (!is_dig(c)) return tok_char(lx, '{');
    do {
        from = from * 10 + (c - '0');
        if (from > 100000) from = 100000;
        c = *++q;
    } while (is_dig(c));

    if (c == '}') {
        to = from;
    } else {
        if (c != ',') return tok_char(lx, '{');
        c = *++q;
        if (c == '}') {
            to = -1;
        } else {
            if (!is_dig(c)) return tok_char(lx, '{');
            to = 0;
            do {
                to = to * 10 + (c - '0');
                if (to > 100000) to = 100000;
                c = *++q;
            } while (is_dig(c));
            if (c != '}') return tok_char(lx, '{');
        }
    }
    lx->p = q + 1;

    if (from >= 500 || to >= 500) return T_BAD;
    if (to >= 0 && from > to) return T_BAD;
    if (from == 0 && to == 1) return '?';
    if (from == 0 && to == -1) return '*';
    if (from == 1 && to == -1) return '+';
    lx->qfrom = (int) from;
    lx->qto = (int) to;
    return T_CQUANT;
}

static void
advance(sre_lex_t *lx)
{
    unsigned c;

    lx->tok_pos = lx->p;
    c = *lx->p;
    if (c == '\0') {
        lx->tok = T_EOF;
        return;
    }
    lx->p++;
    if (strchr("|*+?():.^$", (int) c)) {
        lx->tok = (int) c;
    } else if (c == '\\') {
        lx->tok = lex_escape(lx);
    } else if (c == '[') {
        lx->tok = lex_bracket(lx);
    } else if (c == '{') {
        lx->tok = lex_brace(lx);
    } else {
        lx->tok = tok_char(lx, c);
    }
    if (lx->oom) {
        lx->tok = T_BAD;
    }
}

/* ------------------------------------------------------------------ parser */

static sre_regex_t *parse_alt(sre_lex_t *lx);

static int
starts_atom(int t)
{
    return t == '(' || t == T_CHAR || t == '.' || t == '^' || t == '$'
           || t == T_ASSERT || t == T_CLASS || t == ':';
}

static sre_regex_t *
literal_node(sre_lex_t *lx, unsigned c)
{
    sre_regex_t *n;

    if ((lx->flags & SRE_REGEX_CASELESS)
        && ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')))
    {
        /* a letter becomes the two-member class [c][C] (sre_yyparser.y:244-279) */
        n = node_new(lx, SRE_RE_CLASS, NULL, NULL);
        if (n == NULL) return NULL;
        if (ranges_push(lx, &n->ranges, c, c) != 0) return NULL;
        if (ranges_push(lx, &n->ranges, c ^ 32, c ^ 32) != 0) return NULL;
        return n;
    }
    n = node_new(lx, SRE_RE_LIT, NULL, NULL);
    if (n) n->ch = (uint8_t) c;
    return n;
}

static sre_regex_t *
parse_atom(sre_lex_t *lx)
{
    sre_regex_t *n, *inner;
    int          t = lx->tok;

    switch (t) {
    case '(':
        advance(lx);
        if (lx->tok == '?') {
            advance(lx);
            if (lx->tok != ':') return NULL;
            advance(lx);
            inner = parse_alt(lx);
            if (inner == NULL || lx->tok != ')') return NULL;
            advance(lx);
            return inner;
        }
        {
            /* the group number is taken when '(' is seen (:223-235) */
            sre_uint_t group = ++(*lx->ncaps);
            inner = parse_alt(lx);
            if (inner == NULL || lx->tok != ')') return NULL;
            advance(lx);
            n = node_new(lx, SRE_RE_PAREN, inner, NULL);
            if (n) n->group = group;
            return n;
        }
    case T_CHAR:
        n = literal_node(lx, lx->ch);
        break;
    case ':':
        n = literal_node(lx, ':');
        break;
    case '.':
        n = any_node(lx);
        break;
    case '^':
    case '$':
        n = node_new(lx, SRE_RE_ASSERT, NULL, NULL);
        if (n) n->assertion = (t == '^') ? SRE_ASSERT_CARET : SRE_ASSERT_DOLLAR;
        break;
    case T_ASSERT:
        n = lx->node;
        break;
    case T_CLASS:
        n = lx->node;
        if ((lx->flags & SRE_REGEX_CASELESS) && n->ranges.n
            && ranges_make_caseless(lx, &n->ranges) != 0)
        {
            return NULL;
        }
        break;
    default:
        return NULL;
    }
    if (n == NULL) return NULL;
    advance(lx);
    return n;
}

/* x{n,m}: n copies, then (m-n) copies of x? chained on; x{n,}: n copies then
 * x*.  The copies share one subtree (reference sre_yyparser.y:2011-2084). */
static sre_regex_t *
desugar_counted(sre_lex_t *lx, sre_regex_t *subj, int from, int to, int greedy)
{
    sre_regex_t *acc, *tail;
    int          i;

    if (from == 1 && to == 1) return subj;

    if (from == 0) {
        acc = node_new(lx, SRE_RE_NIL, NULL, NULL);
        i = 0;
    } else {
        acc = subj;
        for (i = 1; acc && i < from; i++) {
            acc = node_new(lx, SRE_RE_CAT, acc, subj);
        }
    }
    if (acc == NULL || from == to) return acc;

    if (to == -1) {
        tail = node_new(lx, SRE_RE_STAR, subj, NULL);
        if (tail == NULL) return NULL;
        tail->greedy = (uint8_t) greedy;
        return node_new(lx, SRE_RE_CAT, acc, tail);
    }

    tail = node_new(lx, SRE_RE_QUEST, subj, NULL);
    if (tail == NULL) return NULL;
    tail->greedy = (uint8_t) greedy;
    for (; acc && i < to; i++) {
        acc = node_new(lx, SRE_RE_CAT, acc, tail);
    }
    return acc;
}

static sre_regex_t *
parse_repeat(sre_lex_t *lx)
{
    sre_regex_t *atom = parse_atom(lx), *n;
    int          t, greedy = 1, from, to;

    if (atom == NULL) return NULL;
    t = lx->tok;
    if (t != '*' && t != '+' && t != '?' && t != T_CQUANT) return atom;

    from = lx->qfrom;
    to = lx->qto;
    advance(lx);
    if (lx->tok == '?') {
        greedy = 0;
        advance(lx);
    }
    if (t == T_CQUANT) {
        return desugar_counted(lx, atom, from, to, greedy);
    }
    n = node_new(lx, t == '*' ? SRE_RE_STAR : t == '+' ? SRE_RE_PLUS : SRE_RE_QUEST,
                 atom, NULL);
    if (n) n->greedy = (uint8_t) greedy;
    return n;
}

static sre_regex_t *
parse_concat(sre_lex_t *lx)
{
    sre_regex_t *acc, *r;

    if (!starts_atom(lx->tok)) {
        return node_new(lx, SRE_RE_NIL, NULL, NULL);
    }
    acc = parse_repeat(lx);
    while (acc && starts_atom(lx->tok)) {
        r = parse_repeat(lx);
        if (r == NULL) return NULL;
        acc = node_new(lx, SRE_RE_CAT, acc, r);
    }
    return acc;
}

static sre_regex_t *
parse_alt(sre_lex_t *lx)
{
    sre_regex_t *acc = parse_concat(lx), *r;

    while (acc && lx->tok == '|') {
        advance(lx);
        r = parse_concat(lx);
        if (r == NULL) return NULL;
        acc = node_new(lx, SRE_RE_ALT, acc, r);
    }
    return acc;
}

/* Parse one regex body; on a syntax error return NULL with *err_offset set to
 * the offending token's offset. */
static sre_regex_t *
parse_one(sre_pool_t *pool, const sre_char *src, sre_uint_t *ncaps, int flags,
    sre_int_t *err_offset)
{
    sre_lex_t    lx;
    sre_regex_t *re;

    memset(&lx, 0, sizeof(lx));
    lx.pool = pool;
    lx.src = lx.p = src ? src : (const uint8_t *) "";
    lx.flags = flags;
    lx.ncaps = ncaps;

    advance(&lx);
    re = parse_alt(&lx);
    if (re == NULL || lx.tok != T_EOF) {
        if (!lx.oom) {
            *err_offset = (sre_int_t) (lx.tok_pos - lx.src);
        }
        return NULL;
    }
    return re;
}

/* ".*?" in front of the alternation of regexes makes every search unanchored
 * leftmost (reference sre_yyparser.y:1830-1857, 1955-1980) */
static sre_regex_t *
wrap_unanchored(sre_lex_t *lx, sre_regex_t *body)
{
    sre_regex_t *dot = node_new(lx, SRE_RE_DOT, NULL, NULL), *star;
    if (dot == NULL) return NULL;
    star = node_new(lx, SRE_RE_STAR, dot, NULL);      /* greedy = 0 */
    if (star == NULL) return NULL;
    return node_new(lx, SRE_RE_CAT, star, body);
}

static sre_regex_t *
wrap_toplevel(sre_lex_t *lx, sre_regex_t *parsed, sre_uint_t group, sre_int_t id)
{
    sre_regex_t *n = node_new(lx, SRE_RE_PAREN, parsed, NULL);    /* $0 */
    if (n == NULL) return NULL;
    n->group = group;
    n = node_new(lx, SRE_RE_TOPLEVEL, n, NULL);
    if (n) n->regex_id = id;
    return n;
}

SRE_API sre_regex_t *
sre_regex_parse(sre_pool_t *pool, sre_char *src, sre_uint_t *ncaps, int flags,
    sre_int_t *err_offset)
{
    sre_lex_t    lx;
    sre_regex_t *re;

    *ncaps = 0;
    *err_offset = -1;
    re = parse_one(pool, src, ncaps, flags, err_offset);
    if (re == NULL) return NULL;

    memset(&lx, 0, sizeof(lx));
    lx.pool = pool;
    re = wrap_toplevel(&lx, re, 0, 0);
    if (re == NULL) return NULL;
    re = wrap_unanchored(&lx, re);
    if (re == NULL) return NULL;
    re->nregexes = 1;
    re->multi_ncaps = sre_palloc(pool, sizeof(sre_uint_t));
    if (re->multi_ncaps == NULL) return NULL;
    re->multi_ncaps[0] = *ncaps;
    return re;
}

SRE_API sre_regex_t *
sre_regex_parse_multi(sre_pool_t *pool, sre_char **regexes, sre_int_t nregexes,
    sre_uint_t *max_ncaps, int *multi_flags, sre_int_t *err_offset,
    sre_int_t *err_regex_id)
{
    sre_lex_t    lx;
    sre_regex_t *all = NULL, *re;
    sre_uint_t   ncaps = 0, base, *multi_ncaps;
    sre_int_t    i;

    *max_ncaps = 0;
    *err_offset = -1;
    *err_regex_id = -1;
    if (nregexes <= 0) return NULL;

    multi_ncaps = sre_palloc(pool, (size_t) nregexes * sizeof(sre_uint_t));
    if (multi_ncaps == NULL) return NULL;
    memset(&lx, 0, sizeof(lx));
    lx.pool = pool;

    /* group numbers run on across regexes: regex i owns groups
     * [base, base + ncaps_i] with `base` its own $0 (:1907-1961) */
    for (i = 0; i < nregexes; i++) {
        *err_regex_id = i;
        base = ncaps;
        re = parse_one(pool, regexes[i], &ncaps, multi_flags ? multi_flags[i] : 0,
                       err_offset);
        if (re == NULL) return NULL;
        re = wrap_toplevel(&lx, re, base, i);
        if (re == NULL) return NULL;
        all = all ? node_new(&lx, SRE_RE_ALT, all, re) : re;
        if (all == NULL) return NULL;
        multi_ncaps[i] = ncaps - base;
        if (multi_ncaps[i] > *max_ncaps) *max_ncaps = multi_ncaps[i];
        ncaps++;
    }

    re = wrap_unanchored(&lx, all);
    if (re == NULL) return NULL;
    re->nregexes = (sre_uint_t) nregexes;
    re->multi_ncaps = multi_ncaps;
    return re;
}

/* ------------------------------------------------------------------ dump */

static void
dump_ranges(const sre_rangevec_t *v)
{
    for (uint32_t i = 0; i < v->n; i++) {
        printf("[%d, %d]", v->r[i].from, v->r[i].to);
    }
}

/* text format pinned by the reference CLI's stdout (reference
 * src/sregex/sre_regex.c:33-167) */
SRE_API void
sre_regex_dump(sre_regex_t *r)
{
    const char *sym;

    switch (r->type) {
    case SRE_RE_ALT:
    case SRE_RE_CAT:
        printf(r->type == SRE_RE_ALT ? "Alt(" : "Cat(");
        sre_regex_dump(r->left);
        printf(", ");
        sre_regex_dump(r->right);
        printf(")");
        break;
    case SRE_RE_LIT:
        printf("Lit(%d)", (int) r->ch);
        break;
    case SRE_RE_DOT:
        printf("Dot");
        break;
    case SRE_RE_NIL:
        printf("Nil");
        break;
    case SRE_RE_PAREN:
        printf("Paren(%lu, ", (unsigned long) r->group);
        sre_regex_dump(r->left);
        printf(")");
        break;
    case SRE_RE_TOPLEVEL:
        printf("TOPLEVEL(%lu, ", (unsigned long) r->regex_id);
        sre_regex_dump(r->left);
        printf(")");
        break;
    case SRE_RE_STAR:
    case SRE_RE_PLUS:
    case SRE_RE_QUEST:
        printf("%s%s(", r->greedy ? "" : "Ng",
               r->type == SRE_RE_STAR ? "Star" : r->type == SRE_RE_PLUS ? "Plus" : "Quest");
        sre_regex_dump(r->left);
        printf(")");
        break;
    case SRE_RE_CLASS:
    case SRE_RE_NCLASS:
        printf(r->type == SRE_RE_CLASS ? "CLASS(" : "NCLASS(");
        dump_ranges(&r->ranges);
        printf(")");
        break;
    case SRE_RE_ASSERT:
        switch (r->assertion) {
        case SRE_ASSERT_BIG_A:   sym = "\\A"; break;
        case SRE_ASSERT_CARET:   sym = "^";   break;
        case SRE_ASSERT_DOLLAR:  sym = "$";   break;
        case SRE_ASSERT_SMALL_Z: sym = "\\z"; break;
        case SRE_ASSERT_BIG_B:   sym = "\\B"; break;
        case SRE_ASSERT_SMALL_B: sym = "\\b"; break;
        default:                 sym = "???"; break;
        }
        printf("ASSERT(%s)", sym);
        break;
    default:
        printf("???");
        break;
    }
}
