// oracle/_ref/libbluenoise.so — a build of the REFERENCE's own header
//     RFW/system/context/rfw/context/blue_noise.h   (tables :5-8203, createBlueNoiseBuffer :8204-8220; includes only <vector>)
// compiled where it lies under /root/reference (oracle/Makefile, target `ref`).  This wrapper contains no reference code: it
// includes the header by the path given on the command line and hands out the table the reference's own function builds.
// Test infrastructure only (tests/, the golden generator); the product never links it.
#include <cstdint>
#include RFW_BLUE_NOISE_HEADER

extern "C" const unsigned int *rfw_ref_blue_noise_table(void)
{
	static const std::vector<unsigned int> table = createBlueNoiseBuffer();
	return table.data();
}
extern "C" unsigned int rfw_ref_blue_noise_words(void) { return 5u * 65536u; }
