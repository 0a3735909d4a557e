// mock: reference-counted pointer with the Teuchos spelling
#ifndef MOCK_TEUCHOS_RCP_HPP
#define MOCK_TEUCHOS_RCP_HPP
#include <memory>
namespace Teuchos {
enum ENull { null };
template <class T>
class RCP {
 public:
  RCP(ENull = null) {}
  explicit RCP(T* p, bool has_ownership = true) { if (has_ownership) p_.reset(p); else p_ = std::shared_ptr<T>(p, [](T*) {}); }
  template <class U> RCP(const RCP<U>& o) : p_(o.shared()) {}
  explicit RCP(const std::shared_ptr<T>& p) : p_(p) {}
  T* get() const { return p_.get(); }
  T* operator->() const { return p_.get(); }
  T& operator*() const { return *p_; }
  bool is_null() const { return !p_; }
  const std::shared_ptr<T>& shared() const { return p_; }
 private:
  std::shared_ptr<T> p_;
};
template <class T> bool operator==(const RCP<T>& p, ENull) { return p.is_null(); }
template <class T> bool operator!=(const RCP<T>& p, ENull) { return !p.is_null(); }
template <class T> RCP<T> rcp(T* p, bool owns = true) { return RCP<T>(p, owns); }
template <class T2, class T1> RCP<T2> rcp_dynamic_cast(const RCP<T1>& p) { return RCP<T2>(std::dynamic_pointer_cast<T2>(p.shared())); }
}  // namespace Teuchos
#endif
