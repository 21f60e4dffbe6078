#pragma once
// stand-in: mrpt/serialization/CSerializable.h, CArchive.h
#include <mrpt/rtti/CObject.h>
#include <cstdint>
namespace mrpt::serialization
{
class CArchive
{
   public:
    template <class T> CArchive& operator<<(const T& v);
    template <class T> CArchive& operator>>(T& v);
};
class CSerializable : public mrpt::rtti::CObject
{
   protected:
    virtual uint8_t serializeGetVersion() const                    = 0;
    virtual void    serializeTo(CArchive& out) const               = 0;
    virtual void    serializeFrom(CArchive& in, uint8_t version)   = 0;
};
}  // namespace mrpt::serialization
#define DEFINE_SERIALIZABLE(Class, NS)                                   \
    DEFINE_MRPT_OBJECT(Class, NS)                                        \
   protected:                                                            \
    uint8_t serializeGetVersion() const override;                        \
    void    serializeTo(mrpt::serialization::CArchive& out) const override; \
    void    serializeFrom(mrpt::serialization::CArchive& in, uint8_t version) override; \
   private:
#define IMPLEMENTS_SERIALIZABLE(Class, Base, NS) IMPLEMENTS_MRPT_OBJECT(Class, Base, NS)
