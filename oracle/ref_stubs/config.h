/* Hand-written config.h used ONLY to compile the read-only reference tree
 * (/root/reference) as the parity oracle `oracle/_ref/*`.  The reference's
 * autotools build is not runnable here (no autoreconf/yaggo); see
 * oracle/README.md.  HAVE_SSE comes from oracle/Makefile (-DHAVE_SSE=1 -msse2), as the reference's own
 * configure turns it on for x86 (m4/m4-ax_ext.m4:221,303, configure.ac:62-67): the SSE2 hash is what a user's
 * build runs, so it is the CPU baseline of record (results identical to the portable path,
 * unit_tests/test_rectangular_binary_matrix.cc:153-177). */
#ifndef ORACLE_REF_CONFIG_H
#define ORACLE_REF_CONFIG_H
#define HAVE_INT128 1
#define HAVE_NUMERIC_LIMITS128 1
#define HAVE_POSIX_MEMALIGN 1
#define HAVE_EXECINFO_H 1
#define HAVE_SYS_SYSCALL_H 1
#define HAVE_EXT_STDIO_FILEBUF_H 1
#define PACKAGE_STRING "jellyfish 2.3.1"
#endif
