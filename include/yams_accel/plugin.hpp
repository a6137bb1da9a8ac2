// plugin.hpp — host side of the plugin boundary: what AbiPluginLoader::load/getInterface do in
// the reference (src/daemon/resource/abi_plugin_loader.cpp:270-442, 657-681), reduced to the
// calls this accelerator needs: dlopen(RTLD_LAZY|RTLD_LOCAL) (:317), yams_plugin_init(config,
// host_context) (:329-345), manifest read (:366-395), yams_plugin_get_interface (:657-681).
#pragma once
#include <dlfcn.h>

#include <memory>
#include <string>

#include "../yams_mi355x_accel.h"
#include "result.hpp"

namespace yams::accel {

// yams_status_t -> ErrorCode exactly as the reference's model-provider adapter maps it
// (src/daemon/resource/abi_model_provider_adapter.cpp:528-552).
inline ErrorCode mapStatus(yams_status_t st) {
    switch (st) {
        case YAMS_OK: return ErrorCode::Success;
        case YAMS_ERR_INVALID_ARG: return ErrorCode::InvalidArgument;
        case YAMS_ERR_NOT_FOUND: return ErrorCode::NotFound;
        case YAMS_ERR_IO: return ErrorCode::IOError;
        case YAMS_ERR_INTERNAL: return ErrorCode::InternalError;
        case YAMS_ERR_UNSUPPORTED: return ErrorCode::NotImplemented;
        case YAMS_ERR_TIMEOUT: return ErrorCode::Timeout;
        case YAMS_ERR_RESOURCE_EXHAUSTED: return ErrorCode::ResourceExhausted;
        default: return ErrorCode::Unknown;
    }
}

class Plugin {
public:
    // Loads libyams_mi355x_accel.so and initialises it.  The library is never dlclose()d, like the
    // reference's plugin tests (tests/plugins/glint/glint_plugin_catch2_test.cpp:47-78).
    static Result<std::shared_ptr<Plugin>> load(const std::string& path,
                                                const std::string& configJson = "{}") {
        void* h = ::dlopen(path.c_str(), RTLD_LAZY | RTLD_LOCAL);
        if (!h) return Error{ErrorCode::FileNotFound, std::string("dlopen failed: ") + ::dlerror()};
        auto p = std::shared_ptr<Plugin>(new Plugin(h));
        auto abi = reinterpret_cast<int (*)()>(::dlsym(h, "yams_plugin_get_abi_version"));
        p->init_ = reinterpret_cast<int (*)(const char*, const void*)>(::dlsym(h, "yams_plugin_init"));
        p->getIface_ = reinterpret_cast<int (*)(const char*, uint32_t, void**)>(::dlsym(h, "yams_plugin_get_interface"));
        p->shutdown_ = reinterpret_cast<void (*)()>(::dlsym(h, "yams_plugin_shutdown"));
        auto manifest = reinterpret_cast<const char* (*)()>(::dlsym(h, "yams_plugin_get_manifest_json"));
        if (!abi || !p->init_ || !p->getIface_ || !manifest)
            return Error{ErrorCode::InvalidData, "not a YAMS plugin (missing entry points)"};
        if (abi() != 1) return Error{ErrorCode::NotSupported, "plugin ABI version mismatch"};
        p->manifest_ = manifest();
        const int rc = p->init_(configJson.c_str(), nullptr);
        if (rc != 0) // YAMS_PLUGIN_ERR_INIT_FAILED: no gfx950 device; the host keeps its CPU backends
            return Error{ErrorCode::NotInitialized, "yams_plugin_init failed rc=" + std::to_string(rc)};
        return p;
    }
    template <typename VTable> Result<VTable*> getInterface(const char* id, uint32_t version) {
        void* out = nullptr;
        const int rc = getIface_(id, version, &out);
        if (rc != 0 || !out) return Error{ErrorCode::NotFound, std::string("interface not served: ") + id};
        return static_cast<VTable*>(out);
    }
    const std::string& manifestJson() const { return manifest_; }
    ~Plugin() { if (shutdown_) shutdown_(); }
private:
    explicit Plugin(void* h) : handle_(h) {}
    void* handle_;
    int (*init_)(const char*, const void*) = nullptr;
    int (*getIface_)(const char*, uint32_t, void**) = nullptr;
    void (*shutdown_)() = nullptr;
    std::string manifest_;
};

} // namespace yams::accel
