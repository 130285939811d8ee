// ts/NodeIndex.ts — node-name interning (reference behaviour: lib/parsing/NodeIndex.ts:1-32; SURVEY.md Appendix D):
// names are case-insensitive, the FIRST spelling of a name is the canonical one (`rev`), id 0 is ground and only the
// literal "0" names it; the matrix row of node id n > 0 is n - 1.
export class NodeIndex {
  private ids: Map<string, number>
  rev: string[]

  constructor() {
    this.ids = new Map()
    this.rev = []
    this.getOrCreate("0")
  }

  getOrCreate(name: string): number {
    const spelled = String(name)
    const key = spelled.toUpperCase()
    const known = this.ids.get(key)
    if (known !== undefined) return known
    this.rev.push(spelled)
    this.ids.set(key, this.rev.length - 1)
    return this.rev.length - 1
  }

  get(name: string): number | undefined {
    return this.ids.get(String(name).toUpperCase())
  }

  count(): number {
    return this.rev.length
  }

  matrixIndexOfNode(nodeId: number): number {
    return nodeId === 0 ? -1 : nodeId - 1
  }
}
