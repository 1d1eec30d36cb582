// HipRT.cpp — the rfw::RenderContext plugin in front of the HIP rendercore: the drop-in for RFW/backends/.
//
// Built as "HipRT.so" (no lib prefix: rfw::system dlopens "<cwd>/<name>.so", RFW/system/src/rfw/system.cpp:119-121)
// and exports the two factory symbols of RFW/system/context/rfw/context/export.h:8-15.  Every virtual call forwards
// to one entry point of include/rfwhip.h; C status codes become std::runtime_error, which is how the reference's
// backends report failures across the plugin boundary (context.h:84-91, utils/logger.h:83-95).
//
// Devices: RFWHIP_DEVICES = "0-7" / "0,2,5" (default: RFWHIP_DEVICE or 0) — the plugin always sits on an rfwhip_group: one
// host thread (the reference's render loop, RFW/system/src/rfw/app.cpp:3-26) drives one context per device, every scene call
// is repeated per context, and the strips are gathered over xGMI (RCCL, or RFWHIP_TRANSPORT=peer) into device 0's image.
//
// Frames in flight: RFWHIP_FRAMES_IN_FLIGHT=n (1..4, default 1) — render_frame(k) enqueues frame k and its presentation and
// returns with frame k - n + 1 on the host (rfwhip_group_present_async / _wait): the devices never idle between frames; with 1
// it returns with frame k finished, like the reference's backends.
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

#include <algorithm>
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
		// the launcher's environment selects the devices: "0-7", "0,2,5", or a single RFWHIP_DEVICE
		std::vector<int> devices;
		if (const char *list = std::getenv("RFWHIP_DEVICES"))
			devices = parse_devices(list);
		if (devices.empty())
		{
			const char *dev = std::getenv("RFWHIP_DEVICE");
			devices.push_back(dev ? std::atoi(dev) : 0);
		}
		int transport = RFWHIP_TRANSPORT_AUTO;
		if (const char *t = std::getenv("RFWHIP_TRANSPORT"))
			transport = std::string(t) == "peer" ? RFWHIP_TRANSPORT_PEER : (std::string(t) == "rccl" ? RFWHIP_TRANSPORT_RCCL : RFWHIP_TRANSPORT_AUTO);
		HIPRT_CHECK(rfwhip_group_create(devices.data(), (int)devices.size(), transport, &m_Group));
		for (int i = 0; i < rfwhip_group_size(m_Group); i++)
			m_Cores.push_back(rfwhip_group_context(m_Group, i));
		if (const char *f = std::getenv("RFWHIP_FRAMES_IN_FLIGHT"))
			m_InFlight = std::max(1, std::min(RFWHIP_PRESENT_SLOTS, std::atoi(f)));
		const char *integ = std::getenv("RFWHIP_INTEGRATOR");
		HIPRT_CHECK(rfwhip_group_set_setting(m_Group, "integrator", integ ? integ : "pt"));
		if (m_InFlight >= RFWHIP_PRESENT_SLOTS) // four frames in flight need four sets of wave buffers (the default ring is three)
			HIPRT_CHECK(rfwhip_group_set_setting(m_Group, "ring", "4"));
#ifdef RFWHIP_HAVE_BLUE_NOISE_TABLE
		{
			// what CUDART does at init (CUDART/src/Context.cpp:43-46): primary rays then use blueNoiseSampler
			const std::vector<unsigned int> table = createBlueNoiseBuffer();
			for (rfwhip_context *c : m_Cores)
				HIPRT_CHECK(rfwhip_set_blue_noise(c, table.data(), table.size()));
			HIPRT_CHECK(rfwhip_group_set_setting(m_Group, "sampler", "bluenoise"));
		}
#endif
	}
	~Context() override
	{
		rfwhip_group_destroy(m_Group);
		m_Group = nullptr, m_Cores.clear();
	}

	static std::vector<int> parse_devices(const std::string &list)
	{
		std::vector<int> out;
		size_t pos = 0;
		while (pos < list.size())
		{
			size_t end = list.find(',', pos);
			if (end == std::string::npos)
				end = list.size();
			const std::string item = list.substr(pos, end - pos);
			const size_t dash = item.find('-');
			if (!item.empty())
			{
				const int a = std::atoi(item.c_str()), b = dash == std::string::npos ? a : std::atoi(item.c_str() + dash + 1);
				for (int d = a; d <= b && out.size() < 64; d++)
					out.push_back(d);
			}
			pos = end + 1;
		}
		return out;
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
		HIPRT_CHECK(rfwhip_group_init(m_Group, width, height));
		// a resize frees the pinned host images of the frames in flight: the window of frames starts over
		m_Frame = 0, m_Latest = nullptr;
		m_Host.assign(size_t(width) * height * 4, 0.0f);
	}

	void cleanup() override
	{
		if (m_Group && !m_Cleaned) // idempotent: called from system::unload and again from destroyRenderContext
		{
			HIPRT_CHECK(rfwhip_group_wait(m_Group));
			for (rfwhip_context *c : m_Cores)
				HIPRT_CHECK(rfwhip_cleanup(c));
			m_Cleaned = true;
		}
	}

	void render_frame(const rfw::Camera &camera, rfw::RenderStatus status) override
	{
		static_assert(sizeof(rfw::Camera) >= sizeof(rfwhip_camera), "camera layout");
		rfwhip_camera cam;
		std::memcpy(&cam, &camera, sizeof(cam)); // position .. pixelCount are the first 60 bytes (camera.h:27-37)
		HIPRT_CHECK(rfwhip_group_render(m_Group, &cam, status == rfw::Reset ? RFWHIP_RESET : RFWHIP_CONVERGE));
		const float *image = nullptr;
		if (m_InFlight >= 2)
		{
			// frame k and its way to the host are enqueued; what is handed out is frame k - (n - 1) (the first n - 1 calls wait
			// for frame 0, their own oldest)
			const int n = m_InFlight;
			HIPRT_CHECK(rfwhip_group_present_async(m_Group, (int)(m_Frame % (unsigned)n)));
			const unsigned long long oldest = m_Frame >= (unsigned)(n - 1) ? m_Frame - (unsigned)(n - 1) : 0;
			HIPRT_CHECK(rfwhip_group_present_wait(m_Group, (int)(oldest % (unsigned)n), &image));
			m_Latest = image;
		}
		else
		{
			HIPRT_CHECK(rfwhip_group_wait(m_Group)); // the reference's render_frame returns with the frame finished
			m_Latest = nullptr;
		}
		m_Frame++;
#ifdef RFWHIP_PLUGIN_WITH_GL
		if (m_Target)
		{
			if (!image)
			{
				HIPRT_CHECK(rfwhip_group_read_framebuffer(m_Group, m_Host.data()));
				image = m_Host.data();
			}
			glBindTexture(GL_TEXTURE_2D, m_Target);
			glTexSubImage2D(GL_TEXTURE_2D, 0, 0, 0, m_Width, m_Height, GL_RGBA, GL_FLOAT, image);
		}
#endif
	}

	// headless read-back: the frame render_frame last handed out (frames in flight: frame k - 1, already on the host)
	int read(float *rgba)
	{
		if (m_Latest)
		{
			std::memcpy(rgba, m_Latest, size_t(m_Width) * m_Height * 4 * sizeof(float));
			return RFWHIP_OK;
		}
		return rfwhip_group_read_framebuffer(m_Group, rgba);
	}

	void set_materials(const std::vector<rfw::DeviceMaterial> &materials,
					   const std::vector<rfw::MaterialTexIds> &texDescriptors) override
	{
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_materials(c, reinterpret_cast<const rfwhip_material *>(materials.data()),
										 reinterpret_cast<const rfwhip_material_tex_ids *>(texDescriptors.data()),
										 materials.size()));
	}

	void set_textures(const std::vector<rfw::TextureData> &textures) override
	{
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_textures(c, reinterpret_cast<const rfwhip_texture *>(textures.data()), textures.size()));
	}

	void set_mesh(size_t index, const rfw::Mesh &mesh) override
	{
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_mesh(c, index, reinterpret_cast<const rfwhip_mesh *>(&mesh)));
	}

	void set_instance(size_t i, size_t meshIdx, const glm::mat4 &transform, const glm::mat3 &inverse_transform) override
	{
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_instance(c, i, meshIdx, reinterpret_cast<const float *>(&transform),
										reinterpret_cast<const float *>(&inverse_transform)));
	}

	void set_sky(const std::vector<glm::vec3> &pixels, size_t width, size_t height) override
	{
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_sky(c, reinterpret_cast<const float *>(pixels.data()), width, height));
	}

	void set_lights(rfw::LightCount lightCount, const rfw::DeviceAreaLight *areaLights,
					const rfw::DevicePointLight *pointLights, const rfw::DeviceSpotLight *spotLights,
					const rfw::DeviceDirectionalLight *directionalLights) override
	{
		rfwhip_light_count n;
		std::memcpy(&n, &lightCount, sizeof(n));
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_lights(c, n, reinterpret_cast<const rfwhip_area_light *>(areaLights),
									  reinterpret_cast<const rfwhip_point_light *>(pointLights),
									  reinterpret_cast<const rfwhip_spot_light *>(spotLights),
									  reinterpret_cast<const rfwhip_directional_light *>(directionalLights)));
	}

	void get_probe_results(unsigned int *instanceIndex, unsigned int *primitiveIndex, float *distance) const override
	{
		// the probe pixel belongs to one rank's strips; the others report nothing for it.  (The probe record is read back by a
		// wait; with frames in flight nothing else waits.)
		if (m_InFlight >= 2)
			HIPRT_CHECK(rfwhip_group_wait(m_Group));
		unsigned int inst = 0, prim = 0;
		float dist = 0.0f;
		// only the rank that owns the probe pixel's 8-row strip has a record for it (every context keeps its last valid one:
		// asking the others would hand back the hit of wherever the probe was before)
		{
			const int owner = rfwhip_row_owner((int)m_ProbeY, (int)m_Cores.size()); // rfwhip.h: serpentine strip ownership
			HIPRT_CHECK(rfwhip_get_probe_results(m_Cores[(size_t)owner], &inst, &prim, &dist));
		}
		if (instanceIndex)
			*instanceIndex = inst;
		if (primitiveIndex)
			*primitiveIndex = prim;
		if (distance)
			*distance = dist;
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
		HIPRT_CHECK(rfwhip_group_set_setting(m_Group, setting.name.c_str(), setting.value.c_str()));
	}

	void update() override { HIPRT_CHECK(rfwhip_group_update(m_Group)); }

	void set_probe_index(glm::uvec2 probePos) override
	{
		m_ProbeY = probePos.y;
		for (rfwhip_context *c : m_Cores)
			HIPRT_CHECK(rfwhip_set_probe_index(c, probePos.x, probePos.y));
	}

	rfw::RenderStats get_stats() const override
	{
		// ray counts add up over the ranks; the times are the slowest rank's
		rfwhip_render_stats st;
		if (m_InFlight >= 2)
			HIPRT_CHECK(rfwhip_group_wait(m_Group)); // (the stats are resolved by a wait; with frames in flight nothing else waits)
		HIPRT_CHECK(rfwhip_get_stats(m_Cores[0], &st));
		for (size_t k = 1; k < m_Cores.size(); k++)
		{
			rfwhip_render_stats r;
			HIPRT_CHECK(rfwhip_get_stats(m_Cores[k], &r));
			st.primaryCount += r.primaryCount, st.secondaryCount += r.secondaryCount, st.deepCount += r.deepCount, st.shadowCount += r.shadowCount;
			st.primaryTime = std::max(st.primaryTime, r.primaryTime), st.secondaryTime = std::max(st.secondaryTime, r.secondaryTime);
			st.deepTime = std::max(st.deepTime, r.deepTime), st.shadowTime = std::max(st.shadowTime, r.shadowTime);
			st.shadeTime = std::max(st.shadeTime, r.shadeTime), st.finalizeTime = std::max(st.finalizeTime, r.finalizeTime);
			st.renderTime = std::max(st.renderTime, r.renderTime), st.animationTime = std::max(st.animationTime, r.animationTime);
		}
		rfw::RenderStats out;
		std::memcpy(&out, &st, sizeof(out));
		return out;
	}

	rfwhip_group *group() const { return m_Group; }

  private:
	rfwhip_group *m_Group = nullptr;
	std::vector<rfwhip_context *> m_Cores;
	bool m_Cleaned = false;
	int m_InFlight = 1;
	unsigned long long m_Frame = 0;
	unsigned m_ProbeY = 0; // row of the probe pixel: decides which rank's record get_probe_results reads
	const float *m_Latest = nullptr;
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
	return static_cast<Context *>(ptr)->read(rgba);
}
