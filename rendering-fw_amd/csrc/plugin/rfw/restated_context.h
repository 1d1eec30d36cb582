// restated_context.h — our own restatement of the declarations a RenderContext plugin needs from
// RFW/system/context/rfw/context/{context.h,structs.h,device_structs.h,camera.h} and glm, for building the plugin
// where the reference's headers (and glm, GLEW, half.hpp) are not installed — as in this image.
//
// Nothing here is copied: only names, member ORDER and byte layouts are reproduced, because the Itanium C++ ABI
// makes exactly those the binary contract of the plugin boundary:
//   * rfw::RenderContext: virtual destructor + 17 virtual methods in the declaration order of context.h:77-110
//     (vtable slot order),
//   * PODs passed by pointer/reference: byte-compatible with include/rfwhip_abi.h (static_asserts below),
//   * std::vector / std::shared_ptr / std::string come from the same libstdc++.
// Where the real headers are available, build HipRT.cpp with -DRFWHIP_USE_RFW_HEADERS and they are used instead.
#pragma once
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "rfwhip_abi.h"

typedef unsigned int GLuint;
typedef unsigned int uint;

namespace glm
{
struct vec2
{
	float x, y;
};
struct vec3
{
	float x, y, z;
};
struct vec4
{
	float x, y, z, w;
};
struct uvec2
{
	unsigned int x, y;
};
struct uvec3
{
	unsigned int x, y, z;
};
struct ivec2
{
	int x, y;
};
struct mat3
{
	float m[9]; // column-major
};
struct mat4
{
	float m[16]; // column-major
};
} // namespace glm

namespace rfw
{
namespace utils
{
class window;
}

enum RenderStatus // context.h:19-23
{
	Reset = 0,
	Converge = 1,
};

enum RenderTarget // context.h:27-34
{
	VULKAN_TEXTURE,
	OPENGL_TEXTURE,
	METAL_TEXTURE,
	BUFFER,
	WINDOW
};

struct AvailableRenderSettings // context.h:36-40
{
	std::vector<std::string> settingKeys;
	std::vector<std::vector<std::string>> settingValues;
};

struct RenderSetting // context.h:42-48
{
	RenderSetting(const std::string &key, std::string val) : name(key), value(std::move(val)) {}
	std::string name;
	std::string value;
};

typedef rfwhip_render_stats RenderStats;		  // context.h:50-72 (12 x 4 bytes)
typedef rfwhip_triangle Triangle;				  // structs.h:24-60
typedef rfwhip_material DeviceMaterial;			  // device_structs.h:56-74
typedef rfwhip_material_tex_ids MaterialTexIds;	  // structs.h:163-167
typedef rfwhip_mesh Mesh;						  // structs.h:175-191
typedef rfwhip_texture TextureData;				  // structs.h:193-205
typedef rfwhip_light_count LightCount;			  // structs.h:207-213
typedef rfwhip_area_light DeviceAreaLight;		  // device_structs.h:105-140
typedef rfwhip_point_light DevicePointLight;	  // device_structs.h:142-150
typedef rfwhip_spot_light DeviceSpotLight;		  // device_structs.h:152-165
typedef rfwhip_directional_light DeviceDirectionalLight; // device_structs.h:167-175

// camera.h:17-60 — only the data members matter across the boundary (the plugin reads them; the non-virtual member
// functions live in the host application).
class Camera
{
  public:
	glm::vec3 position;
	glm::vec3 direction;
	float focalDistance;
	float aperture;
	float brightness;
	float contrast;
	float FOV;
	float aspectRatio;
	float clampValue;
	glm::ivec2 pixelCount;
};
static_assert(sizeof(Camera) == sizeof(rfwhip_camera), "rfw::Camera data members are 60 bytes");

// context.h:74-111 — declaration order == vtable order.
class RenderContext
{
  public:
	RenderContext() = default;
	virtual ~RenderContext() = default;

	[[nodiscard]] virtual std::vector<rfw::RenderTarget> get_supported_targets() const = 0;
	virtual void init(std::shared_ptr<rfw::utils::window> &window)
	{
		(void)window;
		throw std::runtime_error("RenderContext does not support given target type.");
	};
	virtual void init(GLuint *glTextureID, uint width, uint height)
	{
		(void)glTextureID, (void)width, (void)height;
		throw std::runtime_error("RenderContext does not support given target type.");
	};
	virtual void cleanup() = 0;
	virtual void render_frame(const rfw::Camera &camera, rfw::RenderStatus status) = 0;
	virtual void set_materials(const std::vector<rfw::DeviceMaterial> &materials,
							   const std::vector<rfw::MaterialTexIds> &texDescriptors) = 0;
	virtual void set_textures(const std::vector<rfw::TextureData> &textures) = 0;
	virtual void set_mesh(size_t index, const rfw::Mesh &mesh) = 0;
	virtual void set_instance(size_t i, size_t meshIdx, const glm::mat4 &transform, const glm::mat3 &inverse_transform) = 0;
	virtual void set_sky(const std::vector<glm::vec3> &pixels, size_t width, size_t height) = 0;
	virtual void set_lights(rfw::LightCount lightCount, const rfw::DeviceAreaLight *areaLights,
							const rfw::DevicePointLight *pointLights, const rfw::DeviceSpotLight *spotLights,
							const rfw::DeviceDirectionalLight *directionalLights) = 0;
	virtual void get_probe_results(unsigned int *instanceIndex, unsigned int *primitiveIndex, float *distance) const = 0;
	virtual rfw::AvailableRenderSettings get_settings() const = 0;
	virtual void set_setting(const rfw::RenderSetting &setting) = 0;
	virtual void update() = 0;
	virtual void set_probe_index(glm::uvec2 probePos) = 0;
	virtual rfw::RenderStats get_stats() const = 0;
};

} // namespace rfw
