/*
 * rfwhip_abi.h — plain-C restatement of the POD structs that cross the RenderContext plugin boundary of
 * MeirBon/rendering-fw.  No glm, no half.hpp: every field is a scalar or a fixed array, the byte layout is
 * identical to the reference's structs (sizes are static-asserted below), so a reference-side shim can pass
 * `reinterpret_cast`ed pointers straight through.
 *
 * Reference layouts restated here (file:line under /root/reference):
 *   rfw::Triangle / DeviceTriangle  RFW/system/context/rfw/context/structs.h:24-60, device_structs.h:22-33   160 B
 *   rfw::Material / DeviceMaterial  structs.h:85-127, device_structs.h:56-74                                 192 B
 *   rfw::MaterialTexIds             structs.h:163-167                                                        44 B
 *   rfw::Mesh                       structs.h:175-191                                                        56 B
 *   rfw::TextureData                structs.h:193-205                                                        32 B
 *   rfw::LightCount + 4 lights      structs.h:207-255                                            16/96/32/48/32 B
 *   rfw::CameraView                 device_structs.h:95-103                                                  56 B
 *   rfw::Camera (data members)      RFW/system/context/rfw/context/camera.h:27-37                            60 B
 *   rfw::RenderStats                RFW/system/context/rfw/context/context.h:50-72                           48 B
 *   rfw::bvh::BVHNode               RFW/system/bvh/include/bvh/bvh_node.h:23-28, src/bvh_node.cpp:10         32 B
 */
#ifndef RFWHIP_ABI_H
#define RFWHIP_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define RFWHIP_STATIC_ASSERT(c, m) static_assert(c, m)
extern "C" {
#else
#define RFWHIP_STATIC_ASSERT(c, m) _Static_assert(c, m)
#endif

/* structs.h:24-60 — per-face shading record, AoS as the host application hands it over. */
typedef struct rfwhip_triangle
{
	float u0, u1, u2;
	int32_t lightTriIdx; /* -1 = not a light; index into the area-light array otherwise (structs.h:37) */
	float v0, v1, v2;
	uint32_t material;
	float vN0[3];
	float Nx;
	float vN1[3];
	float Ny;
	float vN2[3];
	float Nz;
	float T[3];
	float area;
	float B[3];
	float LOD;
	float vertex0[3];
	float dummy1;
	float vertex1[3];
	float dummy2;
	float vertex2[3];
	float dummy3;
} rfwhip_triangle;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_triangle) == 160, "Triangle must be 160 B (structs.h:60)");

/* One 16-byte texture/normal-map descriptor inside Material (structs.h:99-127). */
typedef struct rfwhip_map_desc
{
	int16_t width, height;
	uint16_t uscale, vscale, uoffs, voffs; /* IEEE binary16 bit patterns */
	uint32_t addr;						   /* index into the TextureData array (EmbreeRT/src/Context.cpp:452) */
} rfwhip_map_desc;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_map_desc) == 16, "map descriptor must be 128 bit");

/* Material flag bits (structs.h:67-83). */
enum rfwhip_mat_flag
{
	RFWHIP_MAT_IS_DIELECTRIC = 0,
	RFWHIP_MAT_DIFFUSE_MAP_IS_HDR = 1,
	RFWHIP_MAT_HAS_DIFFUSE_MAP = 2,
	RFWHIP_MAT_HAS_NORMAL_MAP = 3,
	RFWHIP_MAT_HAS_SPECULARITY_MAP = 4,
	RFWHIP_MAT_HAS_ROUGHNESS_MAP = 5,
	RFWHIP_MAT_IS_ANISOTROPIC = 6,
	RFWHIP_MAT_HAS_2ND_NORMAL_MAP = 7,
	RFWHIP_MAT_HAS_3RD_NORMAL_MAP = 8,
	RFWHIP_MAT_HAS_2ND_DIFFUSE_MAP = 9,
	RFWHIP_MAT_HAS_3RD_DIFFUSE_MAP = 10,
	RFWHIP_MAT_HAS_SMOOTH_NORMALS = 11,
	RFWHIP_MAT_HAS_ALPHA = 12,
	RFWHIP_MAT_HAS_ALPHA_MAP = 13
};

/* structs.h:85-127.  parameters[] holds 16 8-bit Disney parameters in the order of bsdf/compat.h:57-72:
 * x: metallic, subsurface, specular, roughness | y: specTint, anisotropic, sheen, sheenTint |
 * z: clearcoat, clearcoatGloss, transmission, eta*0.5 | w: custom0..3  (material_list.cpp:337-340). */
typedef struct rfwhip_material
{
	uint16_t diffuse[3];	   /* binary16 */
	uint16_t transmittance[3]; /* binary16 */
	uint32_t flags;
	uint32_t parameters[4];
	rfwhip_map_desc map[10]; /* tex0-2, nmap0-2, smap, rmap, cmap, amap */
} rfwhip_material;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_material) == 192, "Material must be 192 B (device_structs.h:56-74)");

typedef struct rfwhip_material_tex_ids
{
	int32_t texture[11];
} rfwhip_material_tex_ids;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_material_tex_ids) == 44, "MaterialTexIds must be 44 B");

/* structs.h:175-191.  All pointers are BORROWED for the duration of the call only. */
typedef struct rfwhip_mesh
{
	const float *vertices;			   /* vec4[vertexCount] */
	const float *normals;			   /* vec3[vertexCount] or NULL */
	const float *texCoords;			   /* vec2[vertexCount] or NULL */
	const rfwhip_triangle *triangles; /* [triangleCount] */
	const uint32_t *indices;		   /* uvec3[triangleCount] or NULL (then triangle i = vertices 3i..3i+2) */
	size_t vertexCount;
	size_t triangleCount;
} rfwhip_mesh;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_mesh) == 56, "Mesh must be 56 B");

enum rfwhip_texture_type
{
	RFWHIP_TEX_FLOAT4 = 0,
	RFWHIP_TEX_UINT = 1
};

/* structs.h:193-205.  UINT texels are r | g<<8 | b<<16 | a<<24, decoded with 1/256 (Context.cpp:466-470). */
typedef struct rfwhip_texture
{
	uint32_t type;
	uint32_t width, height, texelCount;
	uint32_t texAddr;
	uint32_t _pad;
	const void *data;
} rfwhip_texture;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_texture) == 32, "TextureData must be 32 B");

typedef struct rfwhip_light_count
{
	uint32_t areaLightCount, pointLightCount, spotLightCount, directionalLightCount;
} rfwhip_light_count;

typedef struct rfwhip_area_light
{
	float position[3];
	float energy;
	float normal[3];
	float area;
	float radiance[3];
	int32_t dummy0;
	float vertex0[3];
	int32_t triIdx;
	float vertex1[3];
	int32_t instIdx;
	float vertex2[3];
	int32_t dummy1;
} rfwhip_area_light;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_area_light) == 96, "AreaLight must be 96 B");

typedef struct rfwhip_point_light
{
	float position[3];
	float energy;
	float radiance[3];
	int32_t dummy;
} rfwhip_point_light;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_point_light) == 32, "PointLight must be 32 B");

typedef struct rfwhip_spot_light
{
	float position[3];
	float cosInner;
	float radiance[3];
	float cosOuter;
	float direction[3];
	float energy;
} rfwhip_spot_light;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_spot_light) == 48, "SpotLight must be 48 B");

typedef struct rfwhip_directional_light
{
	float direction[3];
	float energy;
	float radiance[3];
	int32_t dummy;
} rfwhip_directional_light;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_directional_light) == 32, "DirectionalLight must be 32 B");

/* camera.h:27-37 — the data members of rfw::Camera in declaration order. */
typedef struct rfwhip_camera
{
	float position[3];
	float direction[3]; /* assumed normalised (Camera.cpp:112) */
	float focalDistance;
	float aperture;
	float brightness;
	float contrast;
	float FOV; /* degrees */
	float aspectRatio;
	float clampValue;
	int32_t pixelCount[2];
} rfwhip_camera;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_camera) == 60, "Camera data members are 60 B");

/* device_structs.h:95-103 */
typedef struct rfwhip_camera_view
{
	float pos[3];
	float p1[3];
	float p2[3];
	float p3[3];
	float aperture;
	float spreadAngle;
} rfwhip_camera_view;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_camera_view) == 56, "CameraView must be 56 B");

/* context.h:50-72 — times in milliseconds. */
typedef struct rfwhip_render_stats
{
	float primaryTime;
	uint32_t primaryCount;
	float secondaryTime;
	uint32_t secondaryCount;
	float deepTime;
	uint32_t deepCount;
	float shadowTime;
	uint32_t shadowCount;
	float shadeTime;
	float finalizeTime;
	float animationTime;
	float renderTime;
} rfwhip_render_stats;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_render_stats) == 48, "RenderStats must be 48 B");

/* bvh_node.h:23-28 — BVH2 node: leaf iff count >= 0 (then left_first = first prim), otherwise the two children
 * live at left_first and left_first+1. */
typedef struct rfwhip_bvh_node
{
	float bmin[3];
	float bmax[3];
	int32_t left_first;
	int32_t count;
} rfwhip_bvh_node;
RFWHIP_STATIC_ASSERT(sizeof(rfwhip_bvh_node) == 32, "BVHNode must be 32 B (bvh_node.cpp:10)");

/* context.h:19-23 */
enum rfwhip_render_status
{
	RFWHIP_RESET = 0,
	RFWHIP_CONVERGE = 1
};

/* Known-answer records of rfwhip_kat(): every function reads one record of RFWHIP_KAT_IN floats and writes one of
 * RFWHIP_KAT_OUT floats (integers travel as their bit patterns).  BSDF record: [0..2] colour, [3..5] absorption,
 * [6..8] ShadingData parameters x,y,z (bsdf/compat.h:57-72), [9..11] N, [12..14] wo, [15..17] wi, [18] t,
 * [19] backfacing, [20] r3 (or r0), [21] r4.  Light record: [0..2] I, [3..5] N, [6] r0, [7] r1, [8] light index,
 * [9..11] O. */
enum
{
	RFWHIP_KAT_IN = 24,
	RFWHIP_KAT_OUT = 8
};
enum rfwhip_kat_function
{
	RFWHIP_KAT_BSDF_EVAL = 0,		   /* disney.h:104-185 BSDFEval            -> rgb */
	RFWHIP_KAT_BSDF_PDF = 1,		   /* disney.h:83-101  BSDFPdf             -> pdf */
	RFWHIP_KAT_BSDF_SAMPLE = 2,		   /* disney.h:188-262 BSDFSample, frame by createTangentSpace(N) -> wi, pdf */
	RFWHIP_KAT_TANGENT_SPACE = 3,	   /* tools.h:204-211                      -> T, B */
	RFWHIP_KAT_PACK_NORMAL = 4,		   /* tools.h:10-29                        -> packed bits, unpacked normal */
	RFWHIP_KAT_RANDOM_BARYCENTRICS = 5, /* lights.h:119-157, r0 = [20]          -> barycentrics */
	RFWHIP_KAT_POINT_ON_LIGHT = 6,	   /* lights.h:159-265 on the context's lights -> P, pickProb, lightPdf, colour */
	RFWHIP_KAT_LIGHT_PICK_PROB = 7,	   /* lights.h:83-116                      -> probability */
	RFWHIP_KAT_BLUE_NOISE = 8,		   /* tools.h:163-181, [0..3] = x, y, sample, dimension (ints) -> value */
	RFWHIP_KAT_HASH = 9,			   /* tools.h:218-235, [0] = seed -> WangHash bits, RandomFloat, state bits */
	RFWHIP_KAT_FASTDIV = 11,		   /* the slot -> pixel mapping's division by per-frame constants (rt_types.h: fast_div): [0..7] = four (n, d) pairs, n < 2^31, 0 < d < 2^31 (ints) -> four quotients (ints) */
	RFWHIP_KAT_TEX_WRAP = 12,		   /* the wrap of a texel coordinate (getShadingData.h:33-41: `% width`, `% height`; rt_core.h: tex_wrap): [0..7] = four (x, w) pairs, 0 <= x < 2^31, 1 <= w < 2^31 (ints) -> four remainders (ints) */
	RFWHIP_KAT_HALF_TO_FLOAT = 10	   /* half -> float as the shade kernels read materials (structs.h:88-117): [0..7] = 8 half bit patterns (ints) -> 8 floats */
};

#ifdef __cplusplus
}
#endif
#endif /* RFWHIP_ABI_H */
