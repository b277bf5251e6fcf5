/* b0_names.c — libphylonium_amd_b0.so: the reference's own symbol names for seam B0,
 *   size_t seqcmp(const char *begin, const char *other, size_t length)      libs/seqcmp.h:14
 *   size_t revseqcmp(const char *begin, const char *other, size_t length)   libs/revseqcmp.h:25
 * (and the function-pointer types of libs/seqcmp.h:25, libs/revseqcmp.h:32-33), forwarding to
 * phylo_seqcmp / phylo_revseqcmp of libphylonium_amd.so.  A separate small library so that the main
 * one does not put such short names into every host's global namespace: phylonium links this one in
 * place of its libseqcmp.a / librevseqcmp.a and nothing else changes. */
#include "../../include/phylonium_amd.h"

size_t seqcmp(const char *begin, const char *other, size_t length) { return phylo_seqcmp(begin, other, length); }
size_t revseqcmp(const char *begin, const char *other, size_t length) { return phylo_revseqcmp(begin, other, length); }
