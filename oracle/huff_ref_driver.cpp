// oracle/huff_ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Component oracle for the Huffman code builder of the reference. This is OUR driver; the
// algorithm under test is NOT restated here: the TU instantiates the reference's own header-only
// template in place (compiled with -I/root/reference/include, see oracle/Makefile), i.e.
//   HuffmanEncoder<15,512>::CreateCodes      (/root/reference/include/mscomp/HuffmanEncoder.h:58-127)
//   HuffmanEncoder<15,512>::CreateCodesSlow  (/root/reference/include/mscomp/HuffmanEncoder.h:129-226)
//
// usage: huff_ref fast|slow  < 512 whitespace-separated counts  > 512 lengths on one line
#include <stdio.h>
#include <string.h>
#include <stdint.h>
#include "mscomp/internal.h"
#include "mscomp/HuffmanEncoder.h"

int main(int argc, char** argv)
{
	if (argc != 2) { fprintf(stderr, "usage: %s fast|slow\n", argv[0]); return 2; }
	uint32_t counts[512];
	for (int i = 0; i < 512; ++i) { if (scanf("%u", &counts[i]) != 1) { return 3; } }
	HuffmanEncoder<15, 512> enc;
	const_bytes lens = strcmp(argv[1], "slow") ? enc.CreateCodes(counts) : enc.CreateCodesSlow(counts);
	for (int i = 0; i < 512; ++i) { printf("%u%c", (unsigned)lens[i], i == 511 ? '\n' : ' '); }
	return 0;
}
