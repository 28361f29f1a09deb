/* Test-only stand-in for the un-vendored deps/snowball submodule: the prototypes the reference's src/ext/default.c
 * mentions in its query EXPANDERS (not part of this path; never called through oracle/_ref). */
#ifndef STUB_LIBSTEMMER_H
#define STUB_LIBSTEMMER_H
struct sb_stemmer;
typedef unsigned char sb_symbol;
struct sb_stemmer *sb_stemmer_new(const char *algorithm, const char *charenc);
void sb_stemmer_delete(struct sb_stemmer *stemmer);
const sb_symbol *sb_stemmer_stem(struct sb_stemmer *stemmer, const sb_symbol *word, int size);
int sb_stemmer_length(struct sb_stemmer *stemmer);
const char **sb_stemmer_list(void);
#endif
