// HipRT.cpp — the rfw::RenderContext plugin in front of the HIP rendercore: the drop-in for RFW/backends/.
//
// Built as "HipRT.so" (no lib prefix: rfw::system dlopens "<cwd>/<name>.so", RFW/system/src/rfw/system.cpp:119-121)
// and exports the two factory symbols of RFW/system/context/rfw/context/export.h:8-15.  Every virtual call forwards
// to one entry point of include/rfwhip.h; C status codes become std::runtime_error, which is how the reference's
// backends report failures across the plugin boundary (context.h:84-91, utils/logger.h:83-95).
//
// Headless by default (RenderTarget::BUFFER, context.h:27-34): the GPU box has no OpenGL.  With
// -DRFWHIP_PLUGIN_WITH_GL (needs GLEW, i.e. the reference's own build environment) render_frame also uploads the
// float4 image into the GL texture handed to init(), the same way EmbreeRT presents (EmbreeRT/src/Context.cpp:289-297).
#ifdef RFWHIP_USE_RFW_HEADERS
#include <rfw/context/context.h>
#include <rfw/context/export.h>
#if __has_include(<rfw/context/blue_noise.h>)
#include <rfw/context/blue_noise.h> // createBlueNoiseBuffer(): the sampler table lives in the reference tree
#define RFWHIP_HAVE_BLUE_NOISE_TABLE 1
#endif
#else
#include "rfw/restated_context.h"
#endif
#include "rfwhip.h"

#ifdef RFWHIP_PLUGIN_WITH_GL
#include <GL/glew.h>
#endif

#include <cstdlib>
#include <string>
#include <vector>

namespace
{

[[noreturn]] void fail(const char *what)
{
	throw std::runtime_error(std::string("HipRT: ") + what + ": " + rfwhip_last_error());
}
#define HIPRT_CHECK(call)          \
	do                             \
	{                              \
		if ((call) != RFWHIP_OK)   \
			fail(#call);           \
	} while (0)

class Context final : public rfw::RenderContext
{
  public:
	Context()
	{
		// one process per GPU: the launcher's environment selects the device / strip ownership
		const char *dev = std::getenv("RFWHIP_DEVICE"), *rank = std::getenv("RFWHIP_RANK"), *world = std::getenv("RFWHIP_WORLD");
		HIPRT_CHECK(rfwhip_create(dev ? std::atoi(dev) : 0, rank ? std::atoi(rank) : 0, world ? std::atoi(world) : 1, &m_Core));
		if (const char *integ = std::getenv("RFWHIP_INTEGRATOR"))
			HIPRT_CHECK(rfwhip_set_setting(m_Core, "integrator", integ));
		else
			HIPRT_CHECK(rfwhip_set_setting(m_Core, "integrator", "pt"));
#ifdef RFWHIP_HAVE_BLUE_NOISE_TABLE
		{
			// what CUDART does at init (CUDART/src/Context.cpp:43-46): primary rays then use blueNoiseSampler
			const std::vector<unsigned int> table = createBlueNoiseBuffer();
			HIPRT_CHECK(rfwhip_set_blue_noise(m_Core, table.data(), table.size()));
			HIPRT_CHECK(rfwhip_set_setting(m_Core, "sampler", "bluenoise"));
		}
#endif
	}
	~Context() override
	{
		rfwhip_destroy(m_Core);
		m_Core = nullptr;
	}

	[[nodiscard]] std::vector<rfw::RenderTarget> get_supported_targets() const override
	{
#ifdef RFWHIP_PLUGIN_WITH_GL
		return {rfw::RenderTarget::OPENGL_TEXTURE, rfw::RenderTarget::BUFFER};
#else
		return {rfw::RenderTarget::BUFFER};
#endif
	}

	void init(std::shared_ptr<rfw::utils::window> &) override
	{
		throw std::runtime_error("HipRT: window targets are not supported.");
	}

	void init(GLuint *glTextureID, uint width, uint height) override
	{
		m_Target = glTextureID ? *glTextureID : 0;
		m_Width = width, m_Height = height;
		HIPRT_CHECK(rfwhip_init(m_Core, width, height));
		m_Host.assign(size_t(width) * height * 4, 0.0f);
	}

	void cleanup() override
	{
		if (m_Core)
			HIPRT_CHECK(rfwhip_cleanup(m_Core)); // idempotent: called from system::unload and destroyRenderContext
	}

	void render_frame(const rfw::Camera &camera, rfw::RenderStatus status) override
	{
		static_assert(sizeof(rfw::Camera) >= sizeof(rfwhip_camera), "camera layout");
		rfwhip_camera cam;
		std::memcpy(&cam, &camera, sizeof(cam)); // position .. pixelCount are the first 60 bytes (camera.h:27-37)
		HIPRT_CHECK(rfwhip_render(m_Core, &cam, status == rfw::Reset ? RFWHIP_RESET : RFWHIP_CONVERGE));
		HIPRT_CHECK(rfwhip_wait(m_Core)); // the reference's render_frame returns with the frame finished
#ifdef RFWHIP_PLUGIN_WITH_GL
		if (m_Target)
		{
			HIPRT_CHECK(rfwhip_read_framebuffer(m_Core, m_Host.data()));
			glBindTexture(GL_TEXTURE_2D, m_Target);
			glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, m_Width, m_Height, GL_RGBA, GL_FLOAT, m_Host.data());
		}
#endif
	}

	void set_materials(const std::vector<rfw::DeviceMaterial> &materials,
					   const std::vector<rfw::MaterialTexIds> &texDescriptors) override
	{
		HIPRT_CHECK(rfwhip_set_materials(m_Core, reinterpret_cast<const rfwhip_material *>(materials.data()),
										 reinterpret_cast<const rfwhip_material_tex_ids *>(texDescriptors.data()),
										 materials.size()));
	}

	void set_textures(const std::vector<rfw::TextureData> &textures) override
	{
		HIPRT_CHECK(rfwhip_set_textures(m_Core, reinterpret_cast<const rfwhip_texture *>(textures.data()), textures.size()));
	}

	void set_mesh(size_t index, const rfw::Mesh &mesh) override
	{
		HIPRT_CHECK(rfwhip_set_mesh(m_Core, index, reinterpret_cast<const rfwhip_mesh *>(&mesh)));
	}

	void set_instance(size_t i, size_t meshIdx, const glm::mat4 &transform, const glm::mat3 &inverse_transform) override
	{
		HIPRT_CHECK(rfwhip_set_instance(m_Core, i, meshIdx, reinterpret_cast<const float *>(&transform),
										reinterpret_cast<const float *>(&inverse_transform)));
	}

	void set_sky(const std::vector<glm::vec3> &pixels, size_t width, size_t height) override
	{
		HIPRT_CHECK(rfwhip_set_sky(m_Core, reinterpret_cast<const float *>(pixels.data()), width, height));
	}

	void set_lights(rfw::LightCount lightCount, const rfw::DeviceAreaLight *areaLights,
					const rfw::DevicePointLight *pointLights, const rfw::DeviceSpotLight *spotLights,
					const rfw::DeviceDirectionalLight *directionalLights) override
	{
		rfwhip_light_count n;
		std::memcpy(&n, &lightCount, sizeof(n));
		HIPRT_CHECK(rfwhip_set_lights(m_Core, n, reinterpret_cast<const rfwhip_area_light *>(areaLights),
									  reinterpret_cast<const rfwhip_point_light *>(pointLights),
									  reinterpret_cast<const rfwhip_spot_light *>(spotLights),
									  reinterpret_cast<const rfwhip_directional_light *>(directionalLights)));
	}

	void get_probe_results(unsigned int *instanceIndex, unsigned int *primitiveIndex, float *distance) const override
	{
		HIPRT_CHECK(rfwhip_get_probe_results(m_Core, instanceIndex, primitiveIndex, distance));
	}

	rfw::AvailableRenderSettings get_settings() const override
	{
		rfw::AvailableRenderSettings s;
		s.settingKeys = {"integrator", "jitter", "spp", "max_depth"};
		s.settingValues = {{"pt", "parity"}, {"xor128", "center"}, {"1", "2", "4", "8", "16"}, {"0", "1", "2", "3", "4"}};
		return s;
	}

	void set_setting(const rfw::RenderSetting &setting) override
	{
		HIPRT_CHECK(rfwhip_set_setting(m_Core, setting.name.c_str(), setting.value.c_str()));
	}

	void update() override { HIPRT_CHECK(rfwhip_update(m_Core)); }

	void set_probe_index(glm::uvec2 probePos) override { HIPRT_CHECK(rfwhip_set_probe_index(m_Core, probePos.x, probePos.y)); }

	rfw::RenderStats get_stats() const override
	{
		rfwhip_render_stats st;
		HIPRT_CHECK(rfwhip_get_stats(m_Core, &st));
		rfw::RenderStats out;
		std::memcpy(&out, &st, sizeof(out));
		return out;
	}

	rfwhip_context *core() const { return m_Core; }

  private:
	rfwhip_context *m_Core = nullptr;
	GLuint m_Target = 0;
	uint m_Width = 0, m_Height = 0;
	std::vector<float> m_Host;
};

} // namespace

#define HIPRT_EXPORT extern "C" __attribute__((visibility("default")))

// export.h:14-15
HIPRT_EXPORT rfw::RenderContext *createRenderContext() { return new Context(); }
HIPRT_EXPORT void destroyRenderContext(rfw::RenderContext *ptr)
{
	ptr->cleanup(); // every reference backend does this too (EmbreeRT/src/Context.cpp:30)
	delete ptr;
}

// Headless hosts (no GL texture to look at) read the BUFFER target through this extra symbol.
HIPRT_EXPORT int hiprtReadFramebuffer(rfw::RenderContext *ptr, float *rgba)
{
	return rfwhip_read_framebuffer(static_cast<Context *>(ptr)->core(), rgba);
}
