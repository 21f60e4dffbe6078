#pragma once
// stand-in: mrpt/rtti/CObject.h (runtime class registry macros)
#include <memory>
namespace mrpt::rtti
{
struct TRuntimeClassId
{
    const char* className;
};
class CObject
{
   public:
    virtual ~CObject() = default;
    virtual const TRuntimeClassId* GetRuntimeClass() const;
};
void registerClass(const TRuntimeClassId* c);
}  // namespace mrpt::rtti
#define DEFINE_MRPT_OBJECT(Class, NS)                                   \
   public:                                                              \
    using Ptr = std::shared_ptr<Class>;                                 \
    static const mrpt::rtti::TRuntimeClassId runtimeClassId;            \
    const mrpt::rtti::TRuntimeClassId* GetRuntimeClass() const override; \
    static std::shared_ptr<mrpt::rtti::CObject> CreateObject();         \
   private:
#define DEFINE_VIRTUAL_MRPT_OBJECT(Class, NS)                           \
   public:                                                              \
    using Ptr = std::shared_ptr<Class>;                                 \
    static const mrpt::rtti::TRuntimeClassId runtimeClassId;            \
   private:
#define IMPLEMENTS_MRPT_OBJECT(Class, Base, NS)                                                        \
    const mrpt::rtti::TRuntimeClassId Class::runtimeClassId = {#NS "::" #Class};                         \
    const mrpt::rtti::TRuntimeClassId* Class::GetRuntimeClass() const { return &Class::runtimeClassId; } \
    std::shared_ptr<mrpt::rtti::CObject> Class::CreateObject() { return std::make_shared<Class>(); }
#define CLASS_ID(T) (&T::runtimeClassId)
