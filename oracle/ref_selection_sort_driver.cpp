/*
 * ref_selection_sort_driver.cpp -- builds the REFERENCE's own CPU twin of SelectionSort into
 * oracle/_ref/libref_selection_sort.so.  TEST INFRASTRUCTURE ONLY.
 *
 * tf_ops/grouping/test/selection_sort.cpp is a stand-alone libc program; it is compiled from where it lies under the
 * reference tree (-DREF_SRC=...), nothing of it is copied here.  Its main() is renamed so the function can be called.
 * selection_sort_cpu prints every row and every pick to stdout: the shim points stdout at /dev/null for the call.
 */
#include <cstdio>
#include <fcntl.h>
#include <unistd.h>
#define main votenet_ref_selection_sort_main
#include REF_SRC
#undef main

extern "C" void ref_selection_sort(int b, int n, int m, int k, const float *dist, int *idx, float *val)
{
    fflush(stdout);
    const int saved = dup(1);
    const int nul = open("/dev/null", O_WRONLY);
    dup2(nul, 1);
    selection_sort_cpu(b, n, m, k, dist, idx, val);
    fflush(stdout);
    dup2(saved, 1);
    close(nul);
    close(saved);
}
