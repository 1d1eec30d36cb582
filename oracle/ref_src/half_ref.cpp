// oracle/_ref/libhalfref.so — a build of the REFERENCE's own half-precision type
//     external/half2.1.0/half.hpp   (half_float::half: the type rfw::DeviceMaterial stores its colours, absorption and
//     texture uv scales / offsets in — RFW/system/context/rfw/context/structs.h:9-10,88-117)
// compiled where it lies under /root/reference (oracle/Makefile, target `ref`).  This wrapper contains no reference code: it
// includes the header by the path given on the command line and exposes the conversions the reference's hosts and kernels
// rely on — half -> float is what the EmbreeRT rendercore does when it reads a material (Context.cpp:417-476), float ->
// half what the material packer does (material_list.cpp).  Test infrastructure only; the product never links it.
#include <cstdint>
#include <cstring>
#include RFW_HALF_HEADER

extern "C" float rfw_ref_half_to_float(uint16_t bits)
{
	half_float::half h;
	static_assert(sizeof(h) == 2, "half is two bytes");
	std::memcpy(&h, &bits, 2);
	return static_cast<float>(h);
}
extern "C" uint16_t rfw_ref_float_to_half(float f)
{
	const half_float::half h(f);
	uint16_t bits;
	std::memcpy(&bits, &h, 2);
	return bits;
}
