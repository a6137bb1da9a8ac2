// result.hpp — the value-or-error vocabulary the reference's seams speak
// (include/yams/core/types.h:25-63 ErrorCode, :147-244 Result<T> in the reference), restated so
// that the adapters in this directory compile without the reference tree.  A host that builds
// inside YAMS includes <yams/core/types.h> instead and defines YAMS_ACCEL_USE_HOST_TYPES.
#pragma once
#ifndef YAMS_ACCEL_USE_HOST_TYPES
#include <stdexcept>
#include <string>
#include <utility>
#include <variant>

namespace yams {

using Hash = std::string; // lower-case hex, core/types.h:17

enum class ErrorCode {
    Success = 0, FileNotFound, PermissionDenied, CorruptedData, StorageFull, InvalidArgument,
    NetworkError, DatabaseError, HashMismatch, ChunkNotFound, ManifestInvalid, TransactionFailed,
    OperationCancelled, OperationInProgress, InvalidOperation, InvalidState, InvalidData,
    InternalError, NotFound, NotSupported, CompressionError, Timeout, TransactionAborted,
    ResourceExhausted, SystemShutdown, ValidationError, WriteError, NotInitialized,
    NotImplemented, InvalidPath, ResourceBusy, IOError, SerializationError, DataCorruption,
    RateLimited, Unauthorized, Unknown
};

struct Error {
    ErrorCode code = ErrorCode::Unknown;
    std::string message;
    Error() = default;
    Error(ErrorCode c, std::string m = {}) : code(c), message(std::move(m)) {}
};

template <typename T> class Result {
public:
    Result(T v) : data_(std::move(v)) {}
    Result(Error e) : data_(std::move(e)) {}
    Result(ErrorCode c) : data_(Error{c}) {}
    bool has_value() const noexcept { return std::holds_alternative<T>(data_); }
    explicit operator bool() const noexcept { return has_value(); }
    T& value() & { if (!has_value()) throw std::runtime_error(error().message); return std::get<T>(data_); }
    const T& value() const& { if (!has_value()) throw std::runtime_error(error().message); return std::get<T>(data_); }
    T&& value() && { if (!has_value()) throw std::runtime_error(error().message); return std::get<T>(std::move(data_)); }
    const Error& error() const { return std::get<Error>(data_); }
private:
    std::variant<T, Error> data_;
};

template <> class Result<void> { // holds an Error whose code Success means OK (core/types.h)
public:
    Result() : err_(ErrorCode::Success, {}) {}
    Result(Error e) : err_(std::move(e)) {}
    Result(ErrorCode c) : err_(c, {}) {}
    bool has_value() const noexcept { return err_.code == ErrorCode::Success; }
    explicit operator bool() const noexcept { return has_value(); }
    void value() const { if (!has_value()) throw std::runtime_error(err_.message); }
    const Error& error() const { return err_; }
private:
    Error err_;
};

} // namespace yams
#else
#include <yams/core/types.h> // the host's own ErrorCode / Error / Result<T> / Hash
#endif
