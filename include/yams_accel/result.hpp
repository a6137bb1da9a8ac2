// result.hpp — the value-or-error vocabulary the reference's seams speak
// (include/yams/core/types.h:25-63 ErrorCode, :147-244 Result<T> in the reference), restated so
// that the adapters in this directory compile without the reference tree.  A host that builds
// inside YAMS includes <yams/core/types.h> instead and defines YAMS_ACCEL_USE_HOST_TYPES.
#pragma once
#ifndef YAMS_ACCEL_USE_HOST_TYPES
#include <cstddef>
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

// The message an Error built from a bare code carries (core/types.h:66-144: the interface's vocabulary — callers
// print and compare these): an indexed table in the order of the enum above.
inline const char* errorToString(ErrorCode c) {
    static const char* const kText[] = {
        "Success", "File not found", "Permission denied", "Corrupted data", "Storage full", "Invalid argument",
        "Network error", "Database error", "Hash mismatch", "Chunk not found", "Invalid manifest", "Transaction failed",
        "Operation cancelled", "Operation in progress", "Invalid operation", "Invalid state", "Invalid data",
        "Internal error", "Not found", "Not supported", "Compression error", "Operation timed out", "Transaction aborted",
        "Resource exhausted", "System shutdown", "Validation error", "Write error", "Not initialized",
        "Not implemented", "Invalid path", "Resource busy", "I/O error", "Serialization error", "Data corruption",
        "Rate limited", "Unauthorized", "Unknown error"};
    const auto i = static_cast<size_t>(c);
    return i < sizeof kText / sizeof kText[0] ? kText[i] : "Unknown error";
}

// core/types.h:147-166: a default Error is SUCCESS (Result<void> relies on it); a bare code carries its standard text;
// a bare message is ErrorCode::Unknown; Error compares with ErrorCode from either side.
struct Error {
    ErrorCode code;
    std::string message;
    Error() : code(ErrorCode::Success) {}
    Error(ErrorCode c, std::string m) : code(c), message(std::move(m)) {}
    Error(ErrorCode c) : code(c), message(errorToString(c)) {}
    Error(std::string m) : code(ErrorCode::Unknown), message(std::move(m)) {}
    bool operator==(ErrorCode c) const { return code == c; }
    bool operator!=(ErrorCode c) const { return code != c; }
    friend bool operator==(ErrorCode c, const Error& e) { return e.code == c; }
    friend bool operator!=(ErrorCode c, const Error& e) { return e.code != c; }
};

// core/types.h:169-214: value() on an error and error() on a value both throw std::runtime_error (with these texts);
// a default-constructed Result<T> is an InternalError ("Uninitialized Result").
template <typename T> class Result {
public:
    Result() : data_(Error{ErrorCode::InternalError, "Uninitialized Result"}) {}
    Result(T v) : data_(std::move(v)) {}
    Result(Error e) : data_(std::move(e)) {}
    Result(ErrorCode c) : data_(Error{c}) {}
    bool has_value() const noexcept { return std::holds_alternative<T>(data_); }
    explicit operator bool() const noexcept { return has_value(); }
    T& value() & { need_value(); return std::get<T>(data_); }
    const T& value() const& { need_value(); return std::get<T>(data_); }
    T&& value() && { need_value(); return std::get<T>(std::move(data_)); }
    const Error& error() const {
        if (has_value()) throw std::runtime_error("Result contains value");
        return std::get<Error>(data_);
    }
private:
    void need_value() const { if (!has_value()) throw std::runtime_error("Result contains error"); }
    std::variant<T, Error> data_;
};

template <> class Result<void> { // core/types.h:217-243: an Error whose code is Success means OK
public:
    Result() = default;
    Result(Error e) : err_(std::move(e)) {}
    Result(ErrorCode c) : err_(Error{c}) {}
    bool has_value() const noexcept { return err_.code == ErrorCode::Success; }
    explicit operator bool() const noexcept { return has_value(); }
    void value() const { if (!has_value()) throw std::runtime_error("Result contains error"); }
    const Error& error() const {
        if (has_value()) throw std::runtime_error("Result contains value");
        return err_;
    }
private:
    Error err_;
};

} // namespace yams
#else
#include <yams/core/types.h> // the host's own ErrorCode / Error / Result<T> / Hash
#endif
