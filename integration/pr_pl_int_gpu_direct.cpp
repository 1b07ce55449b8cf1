// pr_pl_int_gpu_direct.cpp -- PearRay-side adapter: the `gpu_direct` integrator plugin on top of libprgpu.so.
//
// NOT part of this repository's build: it is the file a PearRay maintainer adds under src/plugins/ (it includes PearRay's own
// headers, Eigen etc.) and links against libprgpu.so.  It exports `_pr_exports` through PR_PLUGIN_INIT exactly like the stock
// integrators (src/loader/plugin/Plugin.h:28-66; loaded by PluginManager.cpp:35-37,198-206 when the shared object is named
// pr_pl_int_gpu_direct.so).  Everything it calls in libprgpu.so is declared in include/prgpu.h.
//
// How the scene crosses: core interfaces (IEntity, IMaterial) do not expose triangles or closure parameters, so the adapter does not
// walk Scene -- it hands the SAME scene file the host is loading to the library's own loader (prgpu_prc_load_file), which produces
// the flat prgpu_scene_desc, the Hosek-Wilkie tables of `sky` lights included (prgpu_sky_table: SkyModel.cpp:15-56 restated).
// How the frame comes back: results bypass the per-fragment queue (RenderTileSession::pushSpectralFragment is a per-sample virtual
// call).  The XYZ / sample-count / feedback planes, the shading-point AOVs, the online mean / variance and the light path
// expression planes are downloaded into the host's FrameOutputDevice buffers every `lookahead` iterations (an iteration callback,
// RenderContext.h:110) and at onEnd(), so that the host's observers (src/client/ImageUpdateObserver.cpp:41-60, the network and tev
// observers) see the frame grow while the device renders.
// How the device stays busy: the host's loop is one (tile, iteration) at a time with a barrier per iteration (RenderContext.cpp:234-296);
// the device's unit is a LAUNCH of many iterations.  The first onTile of iteration i that finds nothing issued for it queues
// prgpu_render(i, min(i + lookahead, total)) WITHOUT waiting; the host's iterations inside that range only account their samples.  The
// only waits are the frame deliveries above (prgpu_download waits for the stream) -- at most `lookahead` iterations are ever queued
// ahead of what the host has seen, which also bounds what a soft stop (RenderContext::isStopping) still renders.  Progressive
// rendering (RenderSettings::progressive: no sample count is known, maxSampleCount() == 0) uses a lookahead of one iteration.
#include "Environment.h"
#include "Logger.h"
#include "SceneLoadContext.h"
#include "buffer/FrameBuffer.h"
#include "integrator/IIntegrator.h"
#include "integrator/IIntegratorFactory.h"
#include "integrator/IIntegratorPlugin.h"
#include "output/FrameOutputDevice.h"
#include "output/OutputSystem.h"
#include "renderer/RenderContext.h"
#include "renderer/RenderTile.h"
#include "renderer/RenderTileSession.h"

#include <prgpu.h>

#include <atomic>
#include <mutex>

namespace PR {
struct GpuDirectSetup {
	std::filesystem::path SceneFile;						// the .prc being loaded (SceneLoadContext::currentFile)
	prgpu_settings Integrator;								// `direct` parameters (direct.cpp:500-515)
	uint32 Lookahead = 16;									// iterations per device launch / between frame deliveries
};

// flatten(ctx): everything the device needs, from the scene FILE
static prgpu_prc* flatten(const GpuDirectSetup& setup, const RenderSettings& rs)
{
	prgpu_prc_options opt;
	std::memset(&opt, 0, sizeof(opt));
	opt.width		 = rs.filmWidth; // the host's settings win over the file (command line overrides)
	opt.height		 = rs.filmHeight;
	opt.seed		 = rs.seed;
	opt.force_direct = 1; // the file says (integrator :type 'gpu_direct'); render it with the direct path
	prgpu_prc* file	 = nullptr;
	if (prgpu_prc_load_file(setup.SceneFile.generic_string().c_str(), &opt, &file) != PRGPU_OK) {
		PR_LOG(L_ERROR) << "[gpu_direct] " << prgpu_prc_last_error() << std::endl;
		return nullptr;
	}
	return file;
}

class IntGpuDirect;
class IntGpuDirectInstance : public IIntegratorInstance {
public:
	IntGpuDirectInstance(IntGpuDirect* parent)
		: mParent(parent)
	{
	}
	void onTile(RenderTileSession& session) override;

private:
	IntGpuDirect* mParent;
};

class IntGpuDirect : public IIntegrator {
public:
	explicit IntGpuDirect(const GpuDirectSetup& setup)
		: mSetup(setup)
	{
	}
	~IntGpuDirect() override
	{
		prgpu_scene_destroy(mScene);
		prgpu_prc_free(mFile);
	}

	// RenderContext::start calls onInit once, before the threads exist (RenderContext.cpp:103)
	void onInit(RenderContext* ctx) override
	{
		mContext = ctx;
		mFile	 = flatten(mSetup, ctx->settings());
		if (!mFile)
			throw std::runtime_error("gpu_direct: could not load the scene for the GPU backend");
		prgpu_scene_desc desc = *prgpu_prc_desc(mFile);
		// integrator parameters come from the host's parameter group, everything else from the file
		desc.settings.max_ray_depth		 = mSetup.Integrator.max_ray_depth;
		desc.settings.soft_max_ray_depth = mSetup.Integrator.soft_max_ray_depth;
		desc.settings.mis				 = mSetup.Integrator.mis;
		desc.settings.nee				 = mSetup.Integrator.nee;
		desc.settings.direct			 = mSetup.Integrator.direct;
		desc.settings.emissive_scatter	 = mSetup.Integrator.emissive_scatter;
		if (prgpu_scene_create(&desc, /*device*/ 0, &mScene) != PRGPU_OK)
			throw std::runtime_error(prgpu_last_error()); // same policy as Scene.cpp:103-104
		uint32 n = 0;
		if (const prgpu_output_channel* ch = prgpu_prc_outputs(mFile, &n))
			prgpu_outputs_enable(mScene, ch, n); // the AOV planes the (output ...) blocks ask for
		mIssuedUntil = 0;
		mTotal		 = ctx->settings().maxSampleCount(); // 0: progressive, no end known (RenderSettings.cpp:76-88)
		mLookahead	 = mTotal == 0 ? 1u : std::max<uint32>(1, mSetup.Lookahead);
		// deliver the frame every `lookahead` iterations: called by the last thread of an iteration (RenderContext.cpp:273-296)
		ctx->addIterationCallback([this](const RenderIteration& it) {
			if (it.Pass == 0 && it.Iteration % mLookahead == 0)
				deliver();
		});
	}

	void onEnd() override { deliver(); }

	// the frame so far -> the host's frame buffers (FrameBuffer.h:98-147, [pixel*3+c]); waits for the launches queued so far
	void deliver()
	{
		if (!mScene || !mContext)
			return;
		std::lock_guard<std::mutex> guard(mDeviceMutex);
		for (const auto& dev : mContext->output()->outputDevices()) {
			auto frame = std::dynamic_pointer_cast<FrameOutputDevice>(dev);
			if (!frame)
				continue;
			auto xyz	  = frame->data().getInternalChannel_Spectral(AOV_Output);
			auto samples  = frame->data().getInternalChannel_Counter(AOV_SampleCount);
			auto feedback = frame->data().getInternalChannel_Counter(AOV_Feedback);
			if (prgpu_download(mScene, xyz ? xyz->ptr() : nullptr, samples ? samples->ptr() : nullptr, feedback ? feedback->ptr() : nullptr) != PRGPU_OK)
				PR_LOG(L_ERROR) << "[gpu_direct] " << prgpu_last_error() << std::endl;
			// the planes the (output ...) blocks asked for, where the host's frame holds the same channel (OutputSpecification::setup)
			static const std::pair<AOV3D, uint32> map3d[] = { { AOV_Position, PRGPU_AOV_POSITION }, { AOV_Normal, PRGPU_AOV_NORMAL }, { AOV_NormalG, PRGPU_AOV_NORMAL_G },
															   { AOV_Tangent, PRGPU_AOV_TANGENT }, { AOV_Bitangent, PRGPU_AOV_BITANGENT }, { AOV_View, PRGPU_AOV_VIEW } };
			static const std::pair<AOV1D, uint32> map1d[] = { { AOV_EntityID, PRGPU_AOV_ENTITY_ID }, { AOV_MaterialID, PRGPU_AOV_MATERIAL_ID },
															   { AOV_EmissionID, PRGPU_AOV_EMISSION_ID }, { AOV_Depth, PRGPU_AOV_DEPTH } };
			for (const auto& m : map3d)
				if (auto plane = frame->data().getInternalChannel_3D(m.first))
					(void)prgpu_download_aov(mScene, m.second, plane->ptr()); // EINVAL when the plane was not enabled on the device: left as it is
			for (const auto& m : map1d)
				if (auto plane = frame->data().getInternalChannel_1D(m.first))
					(void)prgpu_download_aov(mScene, m.second, plane->ptr());
			auto mean = frame->data().getInternalChannel_Spectral(AOV_OnlineMean), variance = frame->data().getInternalChannel_Spectral(AOV_OnlineVariance);
			if (mean || variance)
				(void)prgpu_download_variance(mScene, mean ? mean->ptr() : nullptr, variance ? variance->ptr() : nullptr);
			// light path expression planes: the i-th LPE colour channel of the frame is the i-th distinct expression of the file's
			// (channel :type 'color' :lpe ...) list, the order prgpu_outputs_enable enabled them in (FrameContainer.h:78-86,111)
			for (size_t i = 0; i < frame->data().getLPEChannelCount_Spectral(AOV_Output) && i < PRGPU_LPE_MAX; ++i)
				if (auto plane = frame->data().getLPEChannel_Spectral(AOV_Output, i))
					(void)prgpu_download_lpe(mScene, (uint32)i, plane->ptr());
		}
	}

	std::shared_ptr<IIntegratorInstance> createThreadInstance(RenderContext*, size_t) override
	{
		return std::make_shared<IntGpuDirectInstance>(this);
	}

	// RenderThread::main hands every (tile, iteration) to onTile from N threads (RenderThread.cpp:36-70).  The device renders ALL tiles
	// of MANY iterations in one launch: the first thread that meets an iteration nothing has been issued for queues the next
	// `lookahead` iterations and returns at once; every other call only accounts its samples, which keeps the host's iteration barrier,
	// progress display and stop handling working (RenderContext.cpp:234-296).
	void renderIteration(uint32 iteration)
	{
		if (iteration < mIssuedUntil.load(std::memory_order_acquire))
			return;
		std::lock_guard<std::mutex> guard(mDeviceMutex);
		const uint32 begin = mIssuedUntil.load(std::memory_order_relaxed);
		if (iteration < begin)
			return; // another thread was faster
		uint32 end = begin + mLookahead;
		if (mTotal != 0)
			end = std::min<uint32>(end, std::max<uint32>(mTotal, begin + 1));
		if (prgpu_render(mScene, begin, end) != PRGPU_OK) // asynchronous: queued on the scene's stream
			PR_LOG(L_ERROR) << "[gpu_direct] " << prgpu_last_error() << std::endl;
		mIssuedUntil.store(end, std::memory_order_release);
	}

private:
	const GpuDirectSetup mSetup;
	RenderContext* mContext = nullptr;
	prgpu_prc* mFile		= nullptr;
	prgpu_scene* mScene		= nullptr;
	std::atomic<uint32> mIssuedUntil{ 0 }; // iterations [0, mIssuedUntil) are queued on (or done by) the device
	uint32 mTotal	  = 0;				   // iterations of the render (0: progressive)
	uint32 mLookahead = 1;
	std::mutex mDeviceMutex; // one thread at a time talks to the scene object
};

void IntGpuDirectInstance::onTile(RenderTileSession& session)
{
	mParent->renderIteration(session.context()->currentIteration().Iteration);
	session.tile()->statistics().add(RenderStatisticEntry::PixelSampleCount, session.tile()->viewSize().area());
}

class IntGpuDirectFactory : public IIntegratorFactory {
public:
	explicit IntGpuDirectFactory(const GpuDirectSetup& setup)
		: mSetup(setup)
	{
	}
	std::shared_ptr<IIntegrator> createInstance() const override { return std::make_shared<IntGpuDirect>(mSetup); }

private:
	const GpuDirectSetup mSetup;
};

class IntGpuDirectPlugin : public IIntegratorPlugin {
public:
	std::shared_ptr<IIntegratorFactory> create(const std::string&, const SceneLoadContext& ctx) override
	{
		const ParameterGroup& p = ctx.parameters();
		GpuDirectSetup setup;
		setup.SceneFile = ctx.currentFile();
		prgpu_settings_default(&setup.Integrator); // the defaults of direct.cpp:34-39
		setup.Integrator.max_ray_depth		= p.getUInt("max_ray_depth", setup.Integrator.max_ray_depth);
		setup.Integrator.soft_max_ray_depth = std::min<uint32>(setup.Integrator.max_ray_depth, p.getUInt("soft_max_ray_depth", setup.Integrator.soft_max_ray_depth));
		setup.Integrator.mis				= p.getString("mis", "balance") == "power" ? PRGPU_MIS_POWER : PRGPU_MIS_BALANCE;
		setup.Integrator.nee				= p.getBool("nee", true);
		setup.Integrator.direct				= p.getBool("direct", true);
		setup.Integrator.emissive_scatter	= p.getBool("emissive_scatter", true);
		setup.Lookahead						= p.getUInt("lookahead", setup.Lookahead);
		return std::make_shared<IntGpuDirectFactory>(setup);
	}

	const std::vector<std::string>& getNames() const override
	{
		static const std::vector<std::string> names({ "gpu_direct" }); // or { "direct", "standard", "default" } to replace the CPU integrator
		return names;
	}

	PluginSpecification specification(const std::string&) const override
	{
		return PluginSpecificationBuilder("GPU Direct Integrator", "The direct integrator on an MI355X through libprgpu")
			.Identifiers(getNames())
			.Inputs()
			.UInt("max_ray_depth", "Maximum ray depth", 64)
			.UInt("soft_max_ray_depth", "Depth after which russian roulette starts", 4)
			.Option("mis", "MIS mode", "balance", { "balance", "power" })
			.Bool("nee", "Next event estimation", true)
			.Bool("direct", "Direct hits of lights", true)
			.Bool("emissive_scatter", "Emissive surfaces scatter", true)
			.UInt("lookahead", "Iterations per device launch and between frame deliveries", 16)
			.Specification()
			.get();
	}
};
} // namespace PR

PR_PLUGIN_INIT(PR::IntGpuDirectPlugin, _PR_PLUGIN_NAME, PR_PLUGIN_VERSION)
