//! UNVERIFIED (never compiled).
use proc_macro::TokenStream;

/// Accepts the reference's `shader!{ ty: "compute", path: "..." }` and generates no items: nothing outside the
/// replaced src/graphics module reads the generated shader modules.
#[proc_macro]
pub fn shader(_input: TokenStream) -> TokenStream {
    TokenStream::new()
}
