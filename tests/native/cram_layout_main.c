#include <stdio.h>
#include <stddef.h>
size_t ref_layout(size_t *o); size_t our_layout(size_t *o);
int main(void) {
    size_t a[96], b[96]; size_t na = ref_layout(a), nb = our_layout(b); int bad = na != nb;
    for (size_t i = 0; i < na && i < nb; i++) if (a[i] != b[i]) { printf("item %zu: reference %zu, ours %zu\n", i, a[i], b[i]); bad = 1; }
    printf(bad ? "LAYOUT MISMATCH\n" : "layout identical (%zu items)\n", na);
    return bad;
}
