/* The host MD5 of the stream objects, timed alone.  A C file of its own so that it is built by the compiler that builds
 * the library's host objects (gcc, -O2): the figure is then the rate of the code the objects run. */
#include <time.h>
#include "../flac_amd/csrc/host_md5.h"

double tool_md5_rate_pcm16(const int16_t *interleaved, size_t nvalues, uint8_t digest[16])
{
	md5_state m; uint8_t scratch[4096]; struct timespec a, b;
	md5_init(&m);
	clock_gettime(CLOCK_MONOTONIC, &a);
	md5_feed_pcm16(&m, interleaved, nvalues, 16, scratch, sizeof scratch);
	md5_final(&m, digest);
	clock_gettime(CLOCK_MONOTONIC, &b);
	return (double)(b.tv_sec - a.tv_sec) + (double)(b.tv_nsec - a.tv_nsec) * 1e-9;
}
