"""Import-compatibility alias: `from segan.models import SEGAN, ...` (the reference's
package name) resolves to the MI355X implementation in `segan_pytorch_amd`."""
