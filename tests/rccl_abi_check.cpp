// Compile-time check (tests/test_capi_symbols.py: g++ -fsyntax-only) that the hand declarations pm_comm.hip binds RCCL
// through (planetmapper_amd/csrc/pm_rccl_abi.h) are the installed <rccl/rccl.h>'s: sizes, constants, and every bound
// prototype with RCCL's enums read as the ints and its handle / id types as the stand-ins the library passes.
#include <rccl/rccl.h>

#include <type_traits>

#include "pm_rccl_abi.h"

// the installed header's types as the hand declarations spell them
template <class T> struct as_declared { using type = T; };
template <> struct as_declared<ncclResult_t> { using type = int; };
template <> struct as_declared<ncclDataType_t> { using type = int; };
template <> struct as_declared<ncclRedOp_t> { using type = int; };
template <> struct as_declared<ncclComm_t> { using type = pm_rccl::Comm; };
template <> struct as_declared<ncclComm_t *> { using type = pm_rccl::Comm *; };
template <> struct as_declared<ncclUniqueId> { using type = pm_rccl::UniqueId; };
template <> struct as_declared<ncclUniqueId *> { using type = pm_rccl::UniqueId *; };
template <class R, class... A> struct as_declared<R (*)(A...)> {
    using type = typename as_declared<R>::type (*)(typename as_declared<A>::type...);
};

static_assert(std::is_enum<ncclResult_t>::value && sizeof(ncclResult_t) == sizeof(int), "ncclResult_t is an int-sized enum");
static_assert(std::is_enum<ncclDataType_t>::value && sizeof(ncclDataType_t) == sizeof(int), "ncclDataType_t is an int-sized enum");
static_assert(std::is_enum<ncclRedOp_t>::value && sizeof(ncclRedOp_t) == sizeof(int), "ncclRedOp_t is an int-sized enum");
static_assert(std::is_pointer<ncclComm_t>::value && sizeof(ncclComm_t) == sizeof(pm_rccl::Comm), "ncclComm_t is a pointer");
static_assert(NCCL_UNIQUE_ID_BYTES == 128 && sizeof(ncclUniqueId) == sizeof(pm_rccl::UniqueId) && alignof(ncclUniqueId) == alignof(pm_rccl::UniqueId),
              "ncclUniqueId is 128 chars");
static_assert(std::is_trivially_copyable<ncclUniqueId>::value && std::is_standard_layout<ncclUniqueId>::value, "ncclUniqueId passes by value like a char array");
static_assert((int)ncclSuccess == pm_rccl::Success && (int)ncclInt32 == pm_rccl::Int32 && (int)ncclFloat64 == pm_rccl::Float64 && (int)ncclSum == pm_rccl::Sum,
              "enum values");

#define CHECK(name, symbol) \
    static_assert(std::is_same<as_declared<decltype(&nccl##name)>::type, pm_rccl::name##_t>::value, symbol " has another prototype in the installed rccl.h");
PM_RCCL_SYMBOLS(CHECK)

int main() { return 0; }
