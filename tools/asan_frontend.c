/* tools/asan_frontend.c — CPU-only AddressSanitizer run of the host front end
 * (sre_pool.c, sre_parser.c, sre_compiler.c): parses and compiles each pattern
 * given on the command line (or a built-in list of size-explosion patterns).
 *   gcc -fsanitize=address,undefined -g -Iinclude -Isregex_amd/csrc tools/asan_frontend.c \
 *       sregex_amd/csrc/sre_pool.c sregex_amd/csrc/sre_parser.c sregex_amd/csrc/sre_compiler.c -o /tmp/asan_fe
 */
#include <sregex/sregex.h>
#include <stdio.h>
#include <string.h>

static const char *builtin[] = {
    "(?:(?:(?:a{499}){499}){499}){35}", "(?:(?:a{499}){499}){100}", "(?:(?:a{499}){499}){60}",
    "(?:(?:(?:(?:){499}){499}){499}){499}", "x(?:(?:(?:(?:(?:){499}){499}){499}){499}){499}y",
    "(a|b)*a(a|b){15}", "[a-z]+@[a-z]+\\.[a-z]+", "(x{3,}|[^y]{0,499}){2}", "\\bfoo$|^bar\\z", NULL
};

int main(int argc, char **argv)
{
    const char **pats = argc > 1 ? (const char **) argv + 1 : builtin;
    for (; *pats; pats++) {
        sre_pool_t *pool = sre_create_pool(1024);
        sre_uint_t  ncaps;
        sre_int_t   err;
        sre_regex_t *re = sre_regex_parse(pool, (sre_char *) *pats, &ncaps, 0, &err);
        sre_program_t *prog = re ? sre_regex_compile(pool, re) : NULL;
        printf("%-40.40s parse %s compile %s\n", *pats, re ? "ok" : "error", prog ? "ok" : "NULL");
        sre_destroy_pool(pool);
    }
    return 0;
}
